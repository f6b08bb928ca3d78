// wbc_k_qpp.hip — the PACKED stand-alone QP kernel: the reference's own plug-in boundary QP(A, b, C, lb, ub, Clb, Cub).solveQP()
// (QP_Wrapper.py:10-53 -> wbc_qp_solve_ls / wbc_qp_solve, include/wbc.h) with SEVERAL problems per wavefront.
//
// The one-per-wavefront kernel (wbc_k_misc.hip) lights n <= 26 of 64 lanes and fetches its data row by row. Here lane = G r + s: problem r of
// the wave, s = variable / constraint row / working-set slot, G = 16 lanes per problem (four per wavefront; n <= 16 and p <= 16) or 32 (two per
// wavefront; n <= 26, p <= 24). Every stage is written for G lanes with per-problem predication, the loops run to the slowest problem of the wave:
//   loads      everything a problem needs is requested up front (A: dense, coalesced over the problem's m n doubles; C, H: row by row, one
//              element per lane), ~100 loads in flight per lane; C is turned through LDS into registers (lane s keeps row s) and never goes
//              back: the constraint a step works on broadcasts its row through a staging vector;
//   H = A'A    on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), operands straight from the staged A; g = -A'b beside it;
//   presolve   variables with lb == ub leave the problem (qp_core's rule: H_kk = 1, g_k = -value; their columns of C are zeroed);
//   factor     Cholesky fused with L y = e_s, unrolled on fixed registers, the pivot column broadcast through a per-problem LDS vector -> J = L^-T;
//   equalities rows with Clb == Cub enter first, in index order, by the add step of the dual method (Householder on J2; never droppable); the
//              right-hand sides ride along as R'y1 = b_e and x_eq = J1 y1 - J2 J2'g comes from the factors (qp_core's formula);
//   dual loop  Goldfarb-Idnani with per-problem state — the loop of the packed tick kernels (wbc_k_sim3p.hip) on general rows: reductions are
//              DPP row butterflies (+ one v_permlane16_swap for G = 32), a value at a problem-dependent lane comes through ds_bpermute;
//              T = R^-1 is kept packed (upper triangle, column by column) and its products run over the wave's largest working set only;
//   LDS        J [PV][LD] + T packed + five vectors per problem: 9.4 KB (G = 32) / 4.2 KB (G = 16) -> 19 / 17 KB per wavefront, eight
//              wavefronts per CU (the register file's limit at 2 per SIMD);
//   refinement one step at the final working set from the caller's unfactored A, b (qp_refine's formula on T = R^-1), where the problem's
//              pivot ratio asks for it (WBC_REFINE_COND). QP(H, g) has no such residual and is not refined, as on the other kernel.
// Same answers as qp_core to rounding (the same tolerances, the same choice of the most violated constraint, the same iteration count as the
// textbook method: tests/test_gpu_parity.py test_qp_parity_random asserts iters equal to the oracle's).
#include "wbc_packed.h"

namespace wbc {

__device__ __forceinline__ double rowpair_sum(double v) {      // v + (v of the lane 16 away inside the 32-lane half): rows 0|1 and 2|3
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double rowpair_min(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return fmin(__hiloint2double(b[0], a[0]), __hiloint2double(b[1], a[1]));
}
template <int G> __device__ __forceinline__ double gsum(double v) { v = rsum16(v); if (G == 32) v = rowpair_sum(v); return v; }
template <int G> __device__ __forceinline__ double gmin(double v) { v = rmin16(v); if (G == 32) v = rowpair_min(v); return v; }
template <int G> __device__ __forceinline__ unsigned gmask(const unsigned long long bal, const int rbase) {
  return (unsigned)((bal >> rbase) & (G == 32 ? 0xFFFFFFFFull : 0xFFFFull));
}

// LDS block of one problem (doubles): J [PV][LD] and T packed [PV (PV + 1) / 2] (first: the staged A, the H tiles, C on its way into registers),
// then the vectors xv, dv, yv, tv [G each] and ev [32] (b of the staged rows -> the broadcast row of C -> the refinement's residual rows)
template <int G, int PV> struct QppLayout {
  static constexpr int NQ = 64 / G;
  static constexpr int LD = (PV == 16) ? 18 : PV;          // row stride: even (16-byte rows), and "lane = row" b128 reads are conflict-free
  static constexpr int PC = (G == 16) ? 16 : PMAX;         // constraint rows the variant takes
  static constexpr int MCH = (G == 16) ? 16 : 32;          // rows of A staged per pass
  static constexpr int KL = MCH * PV / G;                  // loads per lane and pass of A
  static constexpr int NTP = (PV * (PV + 1) / 2 + 1) & ~1; // packed T, even
  static constexpr int OFF_T = PV * LD;
  static constexpr int OFF_V = OFF_T + NTP;
  static constexpr int RAW = OFF_V + 4 * G + 32;
  // the blocks of a wave's problems sit 128 B (two per wave) / 192 B (four per wave) apart modulo the 256-byte bank row
  static constexpr int PHASE = (G == 32) ? 16 : 24;
  static constexpr int BLOCK = ((RAW - PHASE + 31) & ~31) + PHASE;
  static_assert(MCH * PV <= OFF_V && PC * LD <= OFF_V, "a pass of A, and C, fit the matrix blocks");
  static_assert(MCH <= 32 && PV <= 32, "ev holds a pass of b and a row of C");
};

__device__ __forceinline__ constexpr int tri(const int k) { return k * (k + 1) / 2; }   // packed T: T[i][k] (i <= k) at tri(k) + i

#ifdef WBC_ABLATE
#define QSTOP(k, val) do { if (A.dbg_stop == 400 + (k)) { if (live && has_b) A.x[b * n + s] = (val); return; } } while (0)
#else
#define QSTOP(k, val) do { } while (0)
#endif

// WARM: working sets in / out (QP.solveQPHotstart, QP_Wrapper.py:55-73; the words of wbc_qp_kernel: word 0 bit i / 32 + i = variable i at its lower /
// upper bound, word 1 = constraint rows). The carried set's inequalities that the equalities-only minimiser violates or comes close to are taken by add
// steps without a primal step; x and the multipliers are then read off the factors (x = J1 T'b - J2 J2'g, u = T (T'b + J1'g)); while a seeded
// multiplier is negative its slot is dropped and both are read off again; the dual iterations start from the S-pair that leaves (qp_core's warm
// start, tests/gi_variant.py solve_v3, with the rebuild after every drop instead of the incremental step).
// HALF (G = 32, p <= 16): a row of C is split over the two 16-lane halves of its problem — lane s keeps columns 0..13 of row s & 15 when s < 16, columns
// 14..25 when s >= 16 — 28 VGPRs instead of 52, at the register wall of this kernel; a row's value is the two halves' partial sums added across the
// halves (one v_permlane16_swap), each half reads only its part of the broadcast vector.
template <int G, int PV, bool WARM = false, bool HALF = false>
__global__ void __launch_bounds__(64, 2) wbc_qp_packed_kernel(const QpArgs A) {
  typedef QppLayout<G, PV> L;
  constexpr int NQ = L::NQ, LD = L::LD, PC = L::PC, MCH = L::MCH, KL = L::KL, blk = L::BLOCK;
  __shared__ __attribute__((aligned(16))) double lds[NQ * blk];
  int lane = threadIdx.x, n = A.n, p = A.p, m = A.m;
  asm volatile("" : "+v"(lane), "+s"(n), "+s"(p), "+s"(m));
  const int r = lane / G, s = lane % G, rbase = r * G;
  const int sv = s < PV ? s : PV - 1;                      // (lanes beyond the compiled size shadow the last row; they never write)
  const int b_true = blockIdx.x * NQ + r;
  const bool live = b_true < A.B;
  const size_t b = live ? b_true : A.B - 1;                // a group past the end of the batch shadows the last problem and stores nothing
  double* const Q = lds + r * blk;
  double* const J = Q;
  double* const TP = Q + L::OFF_T;
  double* const xv = Q + L::OFF_V;
  double* const dv = xv + G;
  double* const yv = dv + G;
  double* const tv = yv + G;
  double* const ev = tv + G;
  double* const cl = tv;                                   // Cholesky's column broadcast (tv is free until g is staged)
  const bool has_b = s < n, has_r = s < p;

  // ---- loads: all requested before the first use
  double lb = has_b ? (A.lb ? A.lb[b * n + s] : -1e30) : 0.0;
  double ub = has_b ? (A.ub ? A.ub[b * n + s] : 1e30) : 0.0;
  double clb = has_r ? A.Clb[b * p + s] : 0.0;
  double cub = has_r ? A.Cub[b * p + s] : 0.0;
  double g = 0.0;
  constexpr int NCR = HALF ? 14 : PV;                      // columns of C this lane keeps
  const int crw = HALF ? (s & 15) : s;                     // ... of which row
  const int c0 = (HALF && s >= 16) ? 14 : 0;               // ... from which column on
  const bool c_on = crw < p;
  double crow[NCR];                                        // (zero for rows >= p and beyond column PV - 1)
  {
    double creg[PC];
#pragma unroll
    for (int k = 0; k < PC; ++k) creg[k] = (k < p && has_b) ? A.C[(b * p + k) * n + s] : 0.0;
    double areg[KL > PV ? KL : PV];                        // a pass of A (QP(A, b)) or the columns of H (QP(H, g))
    const double* const Ab = A.A + b * (size_t)m * n;
    const double* const bb = A.bvec + b * (size_t)m;
    if (m > 0) {
      const int lim = ((m < MCH) ? m : MCH) * n;
#pragma unroll
      for (int k = 0; k < KL; ++k) areg[k] = (k * G + s < lim) ? Ab[k * G + s] : 0.0;
    } else {
      const double* const Hb = A.H + b * (size_t)n * n;
#pragma unroll
      for (int k = 0; k < PV; ++k) areg[k] = (k < n && has_b) ? Hb[(size_t)k * n + s] : 0.0;
      g = has_b ? A.g[b * n + s] : 0.0;
    }
    // C: element (k, s) arrives on lane s; through LDS (row stride LD) into row registers
#pragma unroll
    for (int k = 0; k < PC; ++k) if (k < p && s < PV) Q[k * LD + s] = creg[k];
    WSYNC();
#pragma unroll
    for (int k = 0; k < NCR; k += 2) {
      const bool in = c_on && (c0 + k < PV);
      const double2a v = lds2(Q + (in ? crw * LD + c0 + k : 0));
      crow[k] = in ? v.x : 0.0; crow[k + 1] = in ? v.y : 0.0;
    }
    WSYNC();
    if (m > 0) {
      // ---- H = A'A, g = -A'b (QP_Wrapper.py:17-18): A staged MCH rows at a time, dense (row stride n), over the matrix blocks
      constexpr int NT = (G == 32) ? 3 : 1;
      v4f64 acc[NQ][NT];
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[qq][t] = v4f64{0, 0, 0, 0};
      const int kq = lane >> 4, c0 = lane & 15;
#pragma unroll 1
      for (int r0 = 0; r0 < m; r0 += MCH) {
        const int mc = (m - r0 < MCH) ? m - r0 : MCH;
        const int lim = mc * n;
        if (r0 > 0) {
#pragma unroll
          for (int k = 0; k < KL; ++k) areg[k] = (k * G + s < lim) ? Ab[(size_t)r0 * n + k * G + s] : 0.0;
        }
        WSYNC();
#pragma unroll
        for (int k = 0; k < KL; ++k) if (k * G + s < lim) Q[k * G + s] = areg[k];
#pragma unroll
        for (int i0 = 0; i0 < MCH; i0 += G) ev[i0 + s] = (i0 + s < mc) ? bb[r0 + i0 + s] : 0.0;
        WSYNC();
        QSTOP(1, Q[s] + ev[s & 15] + crow[0] + lb + ub + clb + cub);     // loads done, A / b staged, C in registers
        if (G == 16 && has_b) {             // (G = 32: b rides as column 31 of the padded operand and A'b falls out of the same matrix products)
          double g2 = 0.0;
          int rr = 0;
          for (; rr + 1 < mc; rr += 2) { g = fma(-Q[rr * n + s], ev[rr], g); g2 = fma(-Q[(rr + 1) * n + s], ev[rr + 1], g2); }
          if (rr < mc) g = fma(-Q[rr * n + s], ev[rr], g);
          g += g2;
        }
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) {
          const double* const At = lds + qq * blk;
#pragma unroll 1
          for (int s4 = 0; s4 < mc; s4 += 4) {
            const int row = s4 + kq;
            const double a0 = (row < mc && c0 < n) ? At[row * n + c0] : 0.0;
            acc[qq][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, acc[qq][0], 0, 0, 0);
            if (G == 32) {
              const double a1 = (row < mc) ? ((16 + c0 < n) ? At[row * n + 16 + c0] : ((c0 == 15) ? At[(int)(ev - Q) + row] : 0.0)) : 0.0;
              acc[qq][NT > 1 ? 1 : 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a1, acc[qq][NT > 1 ? 1 : 0], 0, 0, 0);
              acc[qq][NT > 2 ? 2 : 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, acc[qq][NT > 2 ? 2 : 0], 0, 0, 0);
            }
          }
        }
      }
      WSYNC();
      // D: lane l, register t holds D[(l >> 4) + 4 t][l & 15]
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) {
        double* const Hq = lds + qq * blk;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = kq + 4 * t, col = c0;
          Hq[row * LD + col] = acc[qq][0][t];
          if (G == 32) {
            if (16 + col < PV) { Hq[row * LD + 16 + col] = acc[qq][NT > 1 ? 1 : 0][t]; Hq[(16 + col) * LD + row] = acc[qq][NT > 1 ? 1 : 0][t]; }
            if (16 + row < PV && 16 + col < PV) Hq[(16 + row) * LD + 16 + col] = acc[qq][NT > 2 ? 2 : 0][t];
            if (col == 15) {                // column 31: A'b
              double* const tq = Hq + (int)(tv - Q);
              tq[row] = -acc[qq][NT > 1 ? 1 : 0][t];
              if (16 + row < PV) tq[16 + row] = -acc[qq][NT > 2 ? 2 : 0][t];
            }
          }
        }
      }
      if (G == 32) { WSYNC(); g = has_b ? tv[s] : 0.0; }
    } else {
#pragma unroll
      for (int k = 0; k < PV; ++k) if (s < PV) Q[k * LD + s] = areg[k];
    }
  }
  WSYNC();
  if (m > 0 && live) {
    if (A.H_out && has_b) { double* const o = A.H_out + b * (size_t)n * n + (size_t)s * n; for (int k = 0; k < n; ++k) o[k] = Q[s * LD + k]; }
    if (A.g_out && has_b) A.g_out[b * n + s] = g;
  }
  QSTOP(2, Q[sv * LD + 1] + g);
  int status = WBC_QP_OPTIMAL;
  bool ok = true;                                          // problem still being solved
  bool refused = false;                                    // refused before the first working-set change (NaN bound, H not positive definite): iters = 0
  // a NaN bound would silently drop its constraint (every comparison with it is false): refuse the problem instead
  {
    const bool bad = (has_b && ((lb != lb) || (ub != ub))) || (has_r && ((clb != clb) || (cub != cub)));
    if (gmask<G>(__ballot(bad), rbase)) { status = WBC_QP_NUMERICAL; ok = false; refused = true; }
  }

  // ---- row s of H into registers; presolve: variables with lb == ub are fixed (qp_core: H_kk = 1, g_k = -value, the value's contribution moves into
  // g and the row bounds, their columns leave H and C). A padded variable (n <= s < PV) is a unit row.
  double h[PV];
#pragma unroll
  for (int k = 0; k < PV; k += 2) { const double2a v = lds2(Q + sv * LD + k); h[k] = v.x; h[k + 1] = v.y; }
  const bool fixb = has_b && (lb == ub) && (fabs(lb) < QP_INF);
  const unsigned fixm = gmask<G>(__ballot(fixb), rbase);
  const int nfix = __popc(fixm);
  if (__ballot(fixb)) {
    const double fv = fixb ? lb : 0.0;
    WSYNC();
    yv[s] = fv;
    WSYNC();
    double gs = 0.0, cs = 0.0;
#pragma unroll
    for (int k = 0; k < PV; k += 2) { const double2a f2 = lds2(yv + k); gs = fma(h[k], f2.x, fma(h[k + 1], f2.y, gs)); }
#pragma unroll
    for (int k = 0; k < NCR; k += 2) { const double2a f2 = lds2(yv + c0 + k); cs = fma(crow[k], f2.x, fma(crow[k + 1], f2.y, cs)); }   // (yv has G = 32 entries: c0 + k < 28)
    if (HALF) cs = rowpair_sum(cs);
    g += gs;
    if (has_r) { clb -= cs; cub -= cs; }
#pragma unroll
    for (int k = 0; k < NCR; ++k) crow[k] = ((fixm >> (c0 + k)) & 1u) ? 0.0 : crow[k];
    if (fixb) { g = -fv; lb = -1e30; ub = 1e30; }          // no longer a constraint
  }
  {
    const bool unit = fixb || s >= n;
#pragma unroll
    for (int k = 0; k < PV; ++k) h[k] = unit ? ((k == sv) ? 1.0 : 0.0) : (((fixm >> k) & 1u) ? 0.0 : h[k]);
  }
  // largest diagonal entry: with the smallest Cholesky pivot it tells whether the refinement has anything to repair (WBC_REFINE_COND)
  double hd = 0.0;
#pragma unroll
  for (int k = 0; k < PV; ++k) hd = (k == sv) ? h[k] : hd;
  const double hmax = -gmin<G>(s < PV ? -hd : 0.0);

  // ---- Cholesky H = L L' fused with the substitution L y = e_s, fully unrolled: step j broadcasts the raw column j of every row through cl and
  // touches the entries k > j only (half the multiply-adds and a quarter of the LDS reads of a rotating-register loop; qp_core has the same form now);
  // y ends as row s of J0 = L^-T
  if (PV > 16) {                                           // the rows of C wait in the (free) J block while h, y and the column fill the registers
    WSYNC();
    if (HALF || has_r) {
#pragma unroll
      for (int k = 0; k < NCR; k += 2) sts2(Q + s * LD + k, crow[k], crow[k + 1]);
    }
  }
  double y[PV];
#pragma unroll
  for (int k = 0; k < PV; ++k) y[k] = (k == sv) ? 1.0 : 0.0;
  double pmin = 1.0;
#pragma unroll
  for (int j = 0; j < PV; ++j) {
    WSYNC();
    if (s < PV) cl[s] = h[j];
    WSYNC();
    double cm[PV];
#pragma unroll
    for (int k = j & ~1; k < PV; k += 2) { const double2a v = lds2(cl + k); cm[k] = v.x; cm[k + 1] = v.y; }
    const double pj = cm[j];
    pmin = (pj > 0.0) ? fmin(pmin, pj) : -1.0;            // (a NaN pivot must fail the test below; fmin would drop it)
    const double rinv = rsqrt(pj), ipj = rinv * rinv;
    const double th = h[j] * ipj, ty = y[j] * ipj;
#pragma unroll
    for (int k = j + 1; k < PV; ++k) { h[k] = fma(-th, cm[k], h[k]); y[k] = fma(-ty, cm[k], y[k]); }
    y[j] *= rinv;
  }
  QSTOP(3, y[0] + y[PV - 1] + h[PV - 1]);
  if (ok && !(pmin > 0.0)) { status = WBC_QP_NUMERICAL; ok = false; refused = true; }
  const bool ill = pmin < WBC_REFINE_COND * hmax;          // the refinement has something to repair
  double sq = 0.0;
#pragma unroll
  for (int k = 0; k < PV; ++k) sq = fma(y[k], y[k], sq);
  const double jf2 = gsum<G>(s < PV ? sq : 0.0);
  WSYNC();
  if (PV > 16) {
#pragma unroll
    for (int k = 0; k < NCR; k += 2) { const bool in = HALF || has_r; const double2a v = lds2(Q + (in ? s : 0) * LD + k); crow[k] = in ? v.x : 0.0; crow[k + 1] = in ? v.y : 0.0; }
    WSYNC();
  }
  if (s < PV) {
#pragma unroll
    for (int k = 0; k < PV; k += 2) sts2(J + s * LD + k, y[k], y[k + 1]);
  }
  tv[s] = has_b ? g : 0.0;
  double cn2 = 0.0;                                        // |C_s|^2
  {
    double c2b = 0.0;
#pragma unroll
    for (int k = 0; k < NCR; k += 2) { cn2 = fma(crow[k], crow[k], cn2); c2b = fma(crow[k + 1], crow[k + 1], c2b); }
    cn2 += c2b;
    if (HALF) cn2 = rowpair_sum(cn2);
  }
  WSYNC();

  // ---- dual active-set state (per problem)
  int actm = 0;                                            // bit 0: this lane's bound is in the working set, bit 1: its row; bits 2 / 3: at the upper side
  double u = 0.0;
  int a_code = 0, q = 0, iters = nfix;
  const int max_iter = 10 * (n + p) + 20;
  // d = J'n of constraint (is_row ? row rr_ : bound of variable ip) with sign sgn, on lane s = slot s; a row's normal is broadcast through ev
  auto normal_d = [&](const bool on, const bool is_row, const int rr_, const int ip, const double sgn) -> double {
    double d;
    if (__ballot(on && is_row)) {
      WSYNC();
      if (on && is_row && crw == rr_) {
#pragma unroll
        for (int k = 0; k < NCR; k += 2) sts2(ev + c0 + k, crow[k], crow[k + 1]);
      }
      WSYNC();
    }
    if (is_row) {
      double d0 = 0.0, d1 = 0.0;
#pragma unroll
      for (int i = 0; i < PV; i += 2) {
        const double2a c2 = lds2(ev + i);
        d0 = fma(J[i * LD + sv], c2.x, d0); d1 = fma(J[(i + 1) * LD + sv], c2.y, d1);
      }
      d = (d0 + d1) * sgn;
    } else d = sgn * J[(ip < PV ? ip : 0) * LD + sv];
    return (has_b && on) ? d : 0.0;
  };
  // value of this lane's own row of C at the vector staged in xv
  auto rowval = [&]() -> double {
    double v = 0.0, vb = 0.0;
#pragma unroll
    for (int k = 0; k < NCR; k += 2) { const double2a x2 = lds2(xv + c0 + k); v = fma(crow[k], x2.x, v); vb = fma(crow[k + 1], x2.y, vb); }
    return HALF ? rowpair_sum(v + vb) : v + vb;
  };
  // r = T d1 on the slots (dv = d staged): row s of T against d over the columns s .. q - 1; the loop runs to the wave's largest working set
  auto t_row_times = [&](const double* const vec) -> double {
    int qm = 0;                              // the wave's largest working set (uniform)
#pragma unroll
    for (int g_ = 0; g_ < NQ; ++g_) { const int qg = __builtin_amdgcn_readlane(q, g_ * G); qm = qg > qm ? qg : qm; }
    double rv = 0.0, rvb = 0.0;
    int k = 0;
#pragma unroll 1
    for (; k + 1 < qm; k += 2) {
      const double t0 = TP[tri(k) + s], t1 = TP[tri(k + 1) + s];
      const double2a d2 = lds2(vec + k);
      rv = fma((s <= k && k < q) ? t0 : 0.0, d2.x, rv); rvb = fma((s <= k + 1 && k + 1 < q) ? t1 : 0.0, d2.y, rvb);
    }
    if (k < qm) { const double t0 = TP[tri(k) + s]; rv = fma((s <= k && k < q) ? t0 : 0.0, vec[k], rv); }
    return rv + rvb;
  };
  // drop slot l of the problems `dr`: Givens sequence read off the removed row of T (rare path)
  auto drop_slot = [&](const bool dr, const int l_) {
    const int l = dr ? l_ : 0;
    const int lc = bpermi(a_code, rbase + l) & 255;
    if (dr && s == ((lc >= n) ? lc - n : lc)) actm &= (lc >= n) ? ~10 : ~5;
    WSYNC();
    yv[s] = u; xv[s] = (double)a_code;
    WSYNC();
    if (dr && s >= l && s < q - 1) { u = yv[s + 1]; a_code = (int)xv[s + 1]; }
    if (dr && s == q - 1) { u = 0.0; a_code = 0; }
    const int srow = (s >= l) ? s + 1 : s;                // old row of T feeding new row s
    double tx = (srow <= l) ? TP[tri(l) + srow] : 0.0;
    double jx = J[sv * LD + l];
    double hrun = TP[tri(l) + l];
    const int kend = dr ? q - 1 : 0;    // this problem's rotations run k = l .. q - 2
#pragma unroll 1
    for (int k0 = 0; k0 < PV - 1; ++k0) {
      const bool on = dr && (l + k0 < kend);
      if (!__ballot(on)) break;
      const int k = on ? l + k0 : 0;
      const double tb = TP[tri(k + 1) + l];
      const double nrm2 = fma(hrun, hrun, tb * tb);
      double c_ = 1.0, s_ = 0.0, rho = 0.0;
      if (nrm2 > 0.0) { const double ri = rsqrt(nrm2); c_ = tb * ri; s_ = -hrun * ri; rho = nrm2 * ri; }
      const double ty_ = (on && srow <= k + 1) ? TP[tri(k + 1) + srow] : 0.0;
      const double jy = J[sv * LD + k + 1];
      WSYNC();
      if (on) {
        hrun = rho;
        if (s <= k) TP[tri(k) + s] = fma(c_, tx, s_ * ty_);
        if (has_b) J[s * LD + k] = fma(c_, jx, s_ * jy);
        tx = fma(-s_, tx, c_ * ty_);
        jx = fma(-s_, jx, c_ * jy);
      }
      WSYNC();
    }
    WSYNC();
    if (dr) {
      if (has_b) J[s * LD + q - 1] = jx;
      --q;
    }
    WSYNC();
  };
  // with d staged (dv = d, yv = d restricted to the slots >= q): z = J2 d2, r = T d1, and the add step's dq = d_q; row s of J stays in registers
  // for the add step
  double jr[PV], v2r[HALF ? PV : 2];               // (HALF: the staged d2 stays too)
  struct Zr { double z, rv, dq, jq; };
  auto products = [&](const bool want_r) -> Zr {
    Zr o;
    double z = 0.0, zb = 0.0;
    const int qc = q < PV ? q : PV - 1;
    o.dq = dv[qc];
    o.jq = J[sv * LD + qc];
#pragma unroll
    for (int k = 0; k < PV; k += 2) {
      const double2a j2 = lds2(J + sv * LD + k); const double2a v2 = lds2(yv + k);
      jr[k] = j2.x; jr[k + 1] = j2.y;
      if (HALF) { v2r[HALF ? k : 0] = v2.x; v2r[HALF ? k + 1 : 0] = v2.y; }
      z = fma(j2.x, v2.x, z); zb = fma(j2.y, v2.y, zb);
    }
    z += zb;
    double rv = 0.0;
    if (want_r) rv = t_row_times(dv);       // nothing to do while no problem of the wave holds an active constraint
    if (s >= q) rv = 0.0;
    if (!has_b) z = 0.0;
    o.z = z; o.rv = rv;
    return o;
  };
  // add: Householder P with P d2 = delta e1; J2 <- J2 P; T gets column (-r/delta, 1/delta); the new slot's multiplier is u_new
  auto add_step = [&](const bool add, const double zn, const Zr& zr, const int wc, const bool is_row, const int rr_, const int ip, const double u_new) -> double {
    const double rsz = frsq(zn), sz = zn * rsz;
    const double delta = (zr.dq >= 0.0) ? -sz : sz;
    const double hv = zn - delta * zr.dq;               // v'v / 2
    const double vv = 2.0 * hv;
    const double jq = zr.jq;
    const double w = (zr.z - delta * jq) * ((vv > 0.0) ? frcp(hv) : 0.0);
    if (add && has_b && vv > 0.0) {
      // J2 <- J2 - w v', v = d2 - delta e_q: the sweep runs on d2 alone (yv = d for k >= q, else 0), entry q is then stored with its own term
#pragma unroll
      for (int k = 0; k < PV; k += 2) {
        double2a v2;
        if (HALF) { v2.x = v2r[HALF ? k : 0]; v2.y = v2r[HALF ? k + 1 : 0]; } else v2 = lds2(yv + k);
        sts2(J + s * LD + k, fma(-w, v2.x, jr[k]), fma(-w, v2.y, jr[k + 1]));
      }
      J[s * LD + q] = fma(-w, zr.dq - delta, jq);
    }
    if (add) {
      const double idel = (zr.dq >= 0.0) ? -rsz : rsz;
      if (s < q) TP[tri(q) + s] = -zr.rv * idel;
      if (s == q) { TP[tri(q) + s] = idel; u = u_new; a_code = wc; }
      if (s == (is_row ? rr_ : ip)) actm |= is_row ? (2 | ((wc >> 8) << 3)) : (1 | ((wc >> 8) << 2));
      ++q;
    }
    return delta;
  };

  // ---- equality rows (Clb == Cub), in index order: the add step without a primal step; y1 solves R'y1 = b_e as the columns come
  const bool eq_r = has_r && (clb == cub) && (fabs(clb) < QP_INF);
  double y1 = 0.0;                                         // slot s < q: y1_s
  {
    unsigned pend = ok ? gmask<G>(__ballot(eq_r), rbase) : 0u;
#pragma unroll 1
    for (;;) {
      const bool on = pend != 0u;
      if (!__ballot(on)) break;
      const int idx = on ? __ffs((int)pend) - 1 : 0;
      pend &= pend - 1;
      if (on) ++iters;
      const double b_e = bperm(clb, rbase + idx), np2 = bperm(cn2, rbase + idx);
      WSYNC();
      const double d = normal_d(on, true, idx, 0, 1.0);
      WSYNC();
      dv[s] = d; yv[s] = (s >= q) ? d : 0.0;
      WSYNC();
      const double zn = gsum<G>(s >= q ? d * d : 0.0);
      const double dy = gsum<G>(s < q ? d * y1 : 0.0);
      const Zr zr = products(__ballot(on && q > 0) != 0);
      const bool add = on && (zn > 100.0 * n * EPS2 * jf2 * np2);
      if (on && !add && !(fabs(dy - b_e) <= 1e-9 * fmax(1.0, fabs(b_e)))) { status = WBC_QP_INFEASIBLE; ok = false; pend = 0u; }   // dependent and inconsistent
      if (__ballot(add)) {
        const int qold = q;
        const double delta = add_step(add, zn, zr, n + idx, true, idx, 0, 0.0);
        if (add && s == qold) y1 = (b_e - dy) / delta;
      }
    }
  }
  const int qe = q;                                        // first inequality slot
  // ---- x on the equalities: x = J1 y1 - J2 (J2'g)   (q = 0: the unconstrained minimiser -J J'g)
  double x;
  {
    WSYNC();
    double t = 0.0, tb = 0.0;
#pragma unroll
    for (int i = 0; i < PV; i += 2) { const double2a g2 = lds2(tv + i); t = fma(J[i * LD + sv], g2.x, t); tb = fma(J[(i + 1) * LD + sv], g2.y, tb); }
    t += tb;
    WSYNC();
    dv[s] = (s < q) ? y1 : (has_b ? -t : 0.0);
    WSYNC();
    double xa = 0.0, xb = 0.0;
#pragma unroll
    for (int k = 0; k < PV; k += 2) { const double2a j2 = lds2(J + sv * LD + k); const double2a v2 = lds2(dv + k); xa = fma(j2.x, v2.x, xa); xb = fma(j2.y, v2.y, xb); }
    x = has_b ? xa + xb : 0.0;
  }

  // ================================ warm start ======================================================
  if (WARM && A.ws_in) {
    const unsigned long long w0 = A.ws_in[2 * b], w1 = A.ws_in[2 * b + 1];
    auto bits = [](const unsigned long long w, const int i) -> int { return (int)(((w >> (i & 31)) & 1ull) | (((w >> (32 + (i & 31))) & 1ull) << 1)); };
    int sb = has_b ? bits(w0, s) : 0, sr = has_r ? bits(w1, s) : 0;
    if (sb == 3) sb = 0;
    if (sr == 3) sr = 0;
    // a seed is taken only if the equalities-only minimiser violates it or comes close to it (within 0.25 max(1, |x|_inf): qp_core, solve_v3 `far`)
    WSYNC();
    xv[s] = x;
    WSYNC();
    const double near = 0.25 * fmax(1.0, -gmin<G>(has_b ? -fabs(x) : 0.0));
    const double vr = rowval();
    const double slb = (sb == 2) ? ub - x : x - lb, slr = (sr == 2) ? cub - vr : vr - clb;
    bool pend_b = ok && has_b && ((sb == 1 && lb > -QP_INF) || (sb == 2 && ub < QP_INF)) && (slb <= near);     // (a fixed variable's bounds are infinite by now)
    bool pend_r = ok && has_r && !eq_r && ((sr == 1 && clb > -QP_INF) || (sr == 2 && cub < QP_INF)) && (slr <= near);
    bool seeded = false;
#pragma unroll 1
    for (;;) {                              // one seed per problem and pass: bounds first, then rows, lowest index first
      const unsigned mb = gmask<G>(__ballot(pend_b), rbase), mr = gmask<G>(__ballot(pend_r), rbase);
      const bool seeding = (mb | mr) != 0u;
      if (!__ballot(seeding)) break;
      const bool is_row = mb == 0u;
      const int idx = seeding ? __ffs((int)(is_row ? mr : mb)) - 1 : 0;
      if (seeding && s == idx) { if (is_row) pend_r = false; else pend_b = false; }
      const int side_b = bpermi(sb, rbase + idx), side_r = bpermi(sr, rbase + idx);
      const double n2r = bperm(cn2, rbase + idx);
      const int c_side = ((is_row ? side_r : side_b) == 2) ? 256 : 0;
      const int wc = (is_row ? n + idx : idx) | c_side;
      const double np2 = is_row ? n2r : 1.0;
      WSYNC();
      const double d = normal_d(seeding, is_row, idx, idx, c_side ? -1.0 : 1.0);
      WSYNC();
      dv[s] = d; yv[s] = (s >= q) ? d : 0.0;
      WSYNC();
      const double zn = gsum<G>(s >= q ? d * d : 0.0);
      const Zr zr = products(__ballot(seeding && q > 0) != 0);
      const bool add = seeding && (zn > 100.0 * n * EPS2 * jf2 * np2);      // (a dependent seed is simply not taken)
      if (__ballot(add)) {
        add_step(add, zn, zr, wc, is_row, idx, idx, 0.0);
        if (add) { seeded = true; ++iters; }
      }
    }
    // x, u of the working set from the factors: b_k = the slots' right-hand sides,  y1 = T'b,  x = J1 y1 - J2 J2'g,  u = T (y1 + J1'g)
    auto rebuild = [&](const bool on) {
      const int cc = a_code & 255, sd = (a_code >> 8) & 1;
      const bool slot = s < q, srow = slot && cc >= n;
      const int rr_ = srow ? cc - n : 0, iv = (slot && !srow) ? cc : 0;
      const double bl = bperm(lb, rbase + iv), bu = bperm(ub, rbase + iv), rl = bperm(clb, rbase + rr_), ru = bperm(cub, rbase + rr_);
      const double bs = slot ? (srow ? (sd ? -ru : rl) : (sd ? -bu : bl)) : 0.0;
      WSYNC();
      dv[s] = bs;
      WSYNC();
      double y1n = 0.0;
#pragma unroll 1
      for (int i = 0; i < PV; ++i) {
        if (!__ballot(slot && i <= s)) break;
        const double t_ = TP[tri(s) + (i <= s ? i : 0)];
        y1n = fma((slot && i <= s) ? t_ : 0.0, dv[i], y1n);
      }
      double jg = 0.0, jgb = 0.0;
#pragma unroll
      for (int i = 0; i < PV; i += 2) { const double2a g2 = lds2(tv + i); jg = fma(J[i * LD + sv], g2.x, jg); jgb = fma(J[(i + 1) * LD + sv], g2.y, jgb); }
      jg += jgb;
      WSYNC();
      yv[s] = slot ? y1n : (has_b ? -jg : 0.0);
      dv[s] = slot ? y1n + jg : 0.0;
      WSYNC();
      double xa = 0.0, xb = 0.0;
#pragma unroll
      for (int k = 0; k < PV; k += 2) { const double2a j2 = lds2(J + sv * LD + k); const double2a v2 = lds2(yv + k); xa = fma(j2.x, v2.x, xa); xb = fma(j2.y, v2.y, xb); }
      const double un = t_row_times(dv);
      if (on) { x = has_b ? xa + xb : 0.0; u = (s >= qe && s < q) ? un : 0.0; }
    };
    if (__ballot(seeded)) {
      rebuild(seeded);
      // RESTORATION: while a seeded multiplier is negative the most negative slot is dropped and x, u are read off the factors again
      bool restoring = seeded;
#pragma unroll 1
      for (;;) {
        const double um = gmin<G>((s >= qe && s < q) ? u : 0.0);
        bool rest = restoring && (um < 0.0);
        if (rest && ++iters > max_iter) { status = WBC_QP_MAX_ITER; rest = false; restoring = false; ok = false; }
        if (!__ballot(rest)) break;
        const unsigned lm = gmask<G>(__ballot(rest && s >= qe && s < q && u == um), rbase);
        const int l = lm ? __ffs((int)lm) - 1 : 0;
        drop_slot(rest, l);
        rebuild(rest);
      }
    }
  }

  QSTOP(4, x);
  // ---- dual iterations for the inequalities
  bool searching = ok;
#pragma unroll 1
  for (;;) {
    // most violated inactive inequality of each problem
    WSYNC();
    xv[s] = x;
    WSYNC();
    double best = 0.0; int code = -1;
    double cand_b = 0.0, cand_n2 = 1.0;    // bound value (signed by side) and |normal|^2 of this lane's candidate
    if (has_b && !(actm & 1)) {
      if (lb > -QP_INF) { const double sl = x - lb; if (sl < -1e-9 * fmax(1.0, fabs(lb)) && sl < best) { best = sl; code = s; cand_b = lb; } }
      if (ub < QP_INF) { const double sl = ub - x; if (sl < -1e-9 * fmax(1.0, fabs(ub)) && sl < best) { best = sl; code = s | 256; cand_b = -ub; } }
    }
    if (p > 0) {
      const double v = rowval();
      if (has_r && !(actm & 2) && !eq_r) {
        if (clb > -QP_INF) { const double sl = v - clb; if (sl < -1e-9 * fmax(1.0, fabs(clb)) && sl < best) { best = sl; code = n + s; cand_b = clb; cand_n2 = cn2; } }
        if (cub < QP_INF) { const double sl = cub - v; if (sl < -1e-9 * fmax(1.0, fabs(cub)) && sl < best) { best = sl; code = (n + s) | 256; cand_b = -cub; cand_n2 = cn2; } }
      }
    }
    const double worst = gmin<G>(best);
    if (searching && !(worst < 0.0)) searching = false;               // primal feasible -> this problem is optimal
#ifdef WBC_ABLATE
    if (A.dbg_stop == 405) searching = false;                         // timing cut: one violation scan, no working-set change
#endif
    if (!__ballot(searching)) break;
    const unsigned wm = gmask<G>(__ballot(searching && best == worst), rbase);
    const int wl = __ffs((int)wm) - 1;                                // first lane of the problem holding the worst violation
    const int wsrc = rbase + (wl < 0 ? 0 : wl);
    const int wc = bpermi(code, wsrc);
    const double b_ip = bperm(cand_b, wsrc);
    const double np2 = bperm(cand_n2, wsrc);
    const int ip = wc & 255, ip_side = (wc >> 8) & 1;
    const bool is_row = ip >= n;
    const int rr_ = is_row ? ip - n : 0;
    const double sgn = ip_side ? -1.0 : 1.0;
    double s_ip = worst, u_ip = 0.0;
    bool stepping = searching;              // problem inside the partial-step loop for its constraint
    int drop_l = -1;
#pragma unroll 1
    for (;;) {
      if (stepping && ++iters > max_iter) { status = WBC_QP_MAX_ITER; stepping = false; searching = false; }
      // ---- drop slot l of the problems that ask for it
      if (__ballot(stepping && drop_l >= 0)) {
        const bool dr = stepping && drop_l >= 0;
        drop_slot(dr, drop_l);
        // slack of the constraint being added, at the current x
        xv[s] = x;
        WSYNC();
        const double vrow = rowval();
        const double v_r = bperm(vrow, rbase + rr_), v_b = bperm(x, rbase + (is_row ? 0 : ip));
        const double v = is_row ? v_r : v_b;
        if (dr) { s_ip = sgn * v - b_ip; drop_l = -1; }
      }
      if (!__ballot(stepping)) break;
      // ---- d = J'n, z = J2 d2, r = T d1
      const double d = normal_d(stepping, is_row, rr_, ip, sgn);
      WSYNC();
      dv[s] = d; yv[s] = (s >= q) ? d : 0.0;
      WSYNC();
      const double zn = gsum<G>(s >= q ? d * d : 0.0);
      const Zr zr = products(__ballot(stepping && q > 0) != 0);
      const double z = zr.z, rv = zr.rv;
      const bool have_step = zn > 100.0 * n * EPS2 * jf2 * np2;
      const bool cand = (s >= qe) && (s < q) && (rv > 2.2250738585072014e-308);   // (normal: frcp's estimate of a denormal is inf)
      const double ratio = cand ? u * frcp(rv) : INFINITY;
      const double t1 = gmin<G>(ratio);
      const unsigned lm = gmask<G>(__ballot(cand && ratio == t1), rbase);
      const int l = (t1 < INFINITY) ? __ffs((int)lm) - 1 : -1;
      const double t2 = have_step ? -s_ip * frcp(zn) : INFINITY;
      const double t = fmin(t1, t2);
      if (stepping && !(t < INFINITY)) { status = WBC_QP_INFEASIBLE; stepping = false; searching = false; }
      if (stepping) {
        if (have_step) x = fma(t, z, x);
        u = fma(-t, rv, u);
        u_ip += t;
      }
      const bool add = stepping && have_step && t == t2;
      if (__ballot(add)) {
        add_step(add, zn, zr, wc, is_row, rr_, ip, u_ip);
        if (add) stepping = false;          // this problem goes back to the search
      }
      if (stepping) drop_l = l;             // blocking slot: dropped at the top of the next pass, then the step is retried
    }
  }

  // ---- iterative refinement at the final working set (qp_refine's step on T = R^-1; QP_Wrapper.py:37 numRefinementSteps), QP(A, b) only and
  // where the pivot ratio asks for it:   gneg = A'(b - A x),  u = -T (J'gneg)_1,  r1 = gneg + N'u,  r2_k = b_k - n_k'x,  x += J1 T'r2 + J2 J2'r1
  {
    const bool need = m > 0 && A.refine > 0 && status == WBC_QP_OPTIMAL && ok && ill;
    if (__ballot(need)) {
      const double* const Ab = A.A + b * (size_t)m * n;
      const double* const bb = A.bvec + b * (size_t)m;
      WSYNC();
      xv[s] = has_b ? x : 0.0;
      WSYNC();
      double gneg = 0.0;
#pragma unroll 1
      for (int r0 = 0; r0 < m; r0 += G) {
        const int row = r0 + s;
        double e = 0.0;
        if (row < m) {
          e = bb[row];
          for (int k = 0; k < n; ++k) e = fma(-Ab[(size_t)row * n + k], xv[k], e);
        }
        WSYNC();
        ev[s] = e;
        WSYNC();
        const int mc = (m - r0 < G) ? m - r0 : G;
        if (has_b) for (int rr = 0; rr < mc; ++rr) gneg = fma(Ab[(size_t)(r0 + rr) * n + s], ev[rr], gneg);
      }
      if (!has_b || fixb) gneg = 0.0;
      // slot s: its constraint, side and bound
      const int cc = a_code & 255, sd = (a_code >> 8) & 1;
      const bool slot = s < q, srow = slot && cc >= n;
      const int rr_ = srow ? cc - n : 0, iv = (slot && !srow) ? cc : 0;
      const double sgn = sd ? -1.0 : 1.0;
      const double bl = bperm(lb, rbase + iv), bu = bperm(ub, rbase + iv), rl = bperm(clb, rbase + rr_), ru = bperm(cub, rbase + rr_);
      const double bnd = srow ? (sd ? ru : rl) : (sd ? bu : bl);
      const double vrow = rowval();                          // this lane's own row of C at x
      // (both fetched by every lane before the choice: a ds_bpermute under a divergent branch reads 0 from the lanes the branch switched off)
      const double val_r = bperm(vrow, rbase + rr_), val_b = bperm(x, rbase + iv);
      const double val = srow ? val_r : val_b;
      const double r2 = slot ? sgn * (bnd - val) : 0.0;
      WSYNC();
      tv[s] = gneg; dv[s] = r2;
      WSYNC();
      // w = J'gneg (lane k: column k of J)
      double w = 0.0, wb = 0.0;
#pragma unroll
      for (int i = 0; i < PV; i += 2) { const double2a g2 = lds2(tv + i); w = fma(J[i * LD + sv], g2.x, w); wb = fma(J[(i + 1) * LD + sv], g2.y, wb); }
      w += wb;
      // dy1 = T'r2 (lane k < q: column k of T, entries i <= k, against r2)
      double dy1 = 0.0;
      for (int i = 0; i < PV; ++i) {
        if (!__ballot(slot && i <= s)) break;
        const double t_ = TP[tri(s) + (i <= s ? i : 0)];
        dy1 = fma((slot && i <= s) ? t_ : 0.0, dv[i], dy1);
      }
      WSYNC();
      yv[s] = slot ? w : 0.0;
      WSYNC();
      const double us = slot ? -t_row_times(yv) * sgn : 0.0;     // u = -T w1: signed multiplier of slot s's constraint
      // r1 = gneg + N'u: a bound's multiplier goes to its variable's lane through dv; the rows' normals come one by one through ev
      WSYNC();
      dv[s] = 0.0;
      WSYNC();
      if (slot && !srow) dv[iv] = us;
      WSYNC();
      double r1 = gneg + dv[s];
#pragma unroll 1
      for (int k = 0; k < PV; ++k) {
        const bool on = k < q && ((bpermi(a_code, rbase + k) & 255) >= n);
        if (!__ballot(k < q)) break;
        if (!__ballot(on)) continue;
        const int rk = on ? (bpermi(a_code, rbase + k) & 255) - n : 0;
        const double uk = bperm(us, rbase + k);
        WSYNC();
        if (on && crw == rk) {
#pragma unroll
          for (int i = 0; i < NCR; i += 2) sts2(ev + c0 + i, crow[i], crow[i + 1]);
        }
        WSYNC();
        if (on) r1 = fma(uk, ev[sv], r1);
      }
      if (!has_b || fixb) r1 = 0.0;
      WSYNC();
      tv[s] = r1;
      WSYNC();
      double dy2 = 0.0, dy2b = 0.0;
#pragma unroll
      for (int i = 0; i < PV; i += 2) { const double2a g2 = lds2(tv + i); dy2 = fma(J[i * LD + sv], g2.x, dy2); dy2b = fma(J[(i + 1) * LD + sv], g2.y, dy2b); }
      dy2 += dy2b;
      WSYNC();
      yv[s] = slot ? dy1 : (s < PV ? dy2 : 0.0);
      WSYNC();
      double da = 0.0, db = 0.0;
#pragma unroll
      for (int k = 0; k < PV; k += 2) { const double2a j2 = lds2(J + sv * LD + k); const double2a w2 = lds2(yv + k); da = fma(j2.x, w2.x, da); db = fma(j2.y, w2.y, db); }
      // the correction is small against x; one that is not (> 0.25 max(1, |x|)) or is non-finite is not applied (qp_refine, oracle: same rule)
      const double dxl = (has_b && !fixb) ? da + db : 0.0;
      const double dmax = -gmin<G>(-fabs(dxl)), xmax = fmax(1.0, -gmin<G>(has_b ? -fabs(x) : 0.0));
      const bool nanr = gmask<G>(__ballot(dxl != dxl), rbase) != 0u;
      if (need && !nanr && dmax <= 0.25 * xmax) x += dxl;
    }
  }

  // ---- outputs. A QP that was not solved returns x = 0 (QP_Wrapper.py:50, 71-73: qpOASES does not write the primal vector of an unsolved problem)
  if (status == WBC_QP_OPTIMAL && gmask<G>(__ballot(has_b && !(fabs(x) <= 1.7976931348623157e308)), rbase)) status = WBC_QP_NUMERICAL;
  if (WARM && A.ws_out) {                  // the working set the next call is seeded with (an unsolved QP carries nothing; equalities are not carried)
    const bool opt = status == WBC_QP_OPTIMAL;
    const bool ab = opt && (actm & 1), ar = opt && (actm & 2) && !eq_r;
    const unsigned long long o0 = (unsigned long long)gmask<G>(__ballot(ab && !(actm & 4)), rbase) | ((unsigned long long)gmask<G>(__ballot(ab && (actm & 4)), rbase) << 32);
    const unsigned long long o1 = (unsigned long long)gmask<G>(__ballot(ar && !(actm & 8)), rbase) | ((unsigned long long)gmask<G>(__ballot(ar && (actm & 8)), rbase) << 32);
    if (live && s == 0) { A.ws_out[2 * b] = o0; A.ws_out[2 * b + 1] = o1; }
  }
  if (live) {
    if (has_b) A.x[b * n + s] = (status == WBC_QP_OPTIMAL) ? x : 0.0;
    if (s == 0) {
      if (A.status) A.status[b] = status;
      if (A.iters) A.iters[b] = refused ? 0 : iters;
    }
  }
}

// One translation unit per PART (csrc/Makefile): part 0 holds the cold variants and the launcher, part 1 the hot-start variants.
#ifndef QPP_PART
#define QPP_PART -1
#endif
#if QPP_PART == 0
extern template __global__ void wbc_qp_packed_kernel<16, 16, true, false>(const QpArgs);
extern template __global__ void wbc_qp_packed_kernel<32, NV, true, false>(const QpArgs);
extern template __global__ void wbc_qp_packed_kernel<32, NV, false, true>(const QpArgs);
extern template __global__ void wbc_qp_packed_kernel<32, NV, true, true>(const QpArgs);
#elif QPP_PART == 1
template __global__ void wbc_qp_packed_kernel<16, 16, true, false>(const QpArgs);
template __global__ void wbc_qp_packed_kernel<32, NV, true, false>(const QpArgs);
#elif QPP_PART == 2
template __global__ void wbc_qp_packed_kernel<32, NV, false, true>(const QpArgs);
template __global__ void wbc_qp_packed_kernel<32, NV, true, true>(const QpArgs);
#endif
#if QPP_PART <= 0
template <int G, int PV>
static int launch_qpp(const QpArgs& a, hipStream_t s) {
  typedef QppLayout<G, PV> L;
  const int grid = (a.B + L::NQ - 1) / L::NQ;
  const bool warm = a.ws_in || a.ws_out;
  if (G == 32 && a.p <= 16) {              // rows of C split over the two halves of a problem
    if (warm) hipLaunchKernelGGL((wbc_qp_packed_kernel<G, PV, true, (G == 32)>), dim3(grid), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((wbc_qp_packed_kernel<G, PV, false, (G == 32)>), dim3(grid), dim3(64), 0, s, a);
  } else if (warm) hipLaunchKernelGGL((wbc_qp_packed_kernel<G, PV, true, false>), dim3(grid), dim3(64), 0, s, a);
  else hipLaunchKernelGGL((wbc_qp_packed_kernel<G, PV, false, false>), dim3(grid), dim3(64), 0, s, a);
  return check_launch("qp packed");
}

// 4: four problems per wavefront, 2: two
int qp_packed_lanes(const QpArgs& a) {
  if (a.n <= 16 && a.p <= 16) return 4;
  return 2;
}
int launch_qp_packed(const QpArgs& a, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (qp_packed_lanes(a) == 4) return launch_qpp<16, 16>(a, s);
  return launch_qpp<32, NV>(a, s);
}
#endif

}  // namespace wbc
