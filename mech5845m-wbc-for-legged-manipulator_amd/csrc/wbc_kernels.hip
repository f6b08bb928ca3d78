// wbc_kernels.hip — gfx950 (MI355X, CDNA4) kernels of the batched whole-body-control tick.
//
// One robot instance per 64-lane wavefront, one wavefront per workgroup, one workgroup per instance (grid = B):
//   lane j  <-> joint j          during forward kinematics (level-synchronous over the tree depth),
//   lane k  <-> velocity DoF k   everywhere else (column k of every Jacobian, row/column k of H, J, T).
// All per-instance matrices live in LDS (row stride 26 doubles: 26 ≡ 2 mod 4 makes both the "lane = row,
// ds_read_b128 along the row" and the "lane = column, ds_read_b64 down the column" patterns bank-conflict free on
// the 64-bank LDS of CDNA4); lane-distributed vectors live in VGPRs, wave-uniform scalars in SGPRs. Every loop
// over the matrix dimension is a real loop with a compact body: the whole tick is a few thousand instructions of
// code, so the waves of a CU, each in a different phase, share the 64 KB instruction cache without evicting each
// other (the first, fully unrolled register-resident version was 82 KB of code and instruction-fetch bound:
// profiles/r01_*_v1.*). Wave reductions use DPP row operations + v_readlane, never the LDS crossbar.
// HBM traffic per tick is the instance's own inputs/outputs (~0.7 KB, coalesced).
// Kernels: wbc_tick_kernel<MODE> (general path: tick / assemble / FK outputs), wbc_tick_sim3_kernel (+ wbc_tick_deferred_kernel:
// the benchmark path — contact equalities eliminated structurally, reduced QP assembled directly, compact LDS, 3 waves per
// SIMD), wbc_posture_kernel (MANI/HYBRID posture target), wbc_update_kernel (updateState + trunkWorldPos, roll-out state),
// wbc_qp_kernel (QP(A, b, ...) boundary), wbc_integrate_kernel.
//
// Reference semantics (file:line relative to the reference repo) are cited at each stage; the CPU restatement the
// tests compare against is oracle/wbc_oracle.c (never linked here); the algebra of the QP variant is stated in
// plain numpy in tests/gi_variant.py (solve_v2).
#include <hip/hip_runtime.h>
#include <math.h>
#include "wbc_device.h"

namespace wbc {

constexpr int LDJ = 26;                 // LDS row stride (doubles) of the n x n matrices and of Cm
constexpr int PMAX = WBC_MAX_P;         // 24
constexpr double QP_INF = 1e20;
constexpr double EPS2 = 2.220446049250313e-16 * 2.220446049250313e-16;

// staging image of one instance's inputs (doubles)
constexpr int IN_Q = 0, IN_EET = 28, IN_EEP = 43, IN_BOX = 58;                       // group 1 (lanes 0..61)
constexpr int IN_TT = 64, IN_TP = 67, IN_TRE = 70, IN_TPR = 73, IN_CT = 82, IN_CV = 85;  // group 2 (24 values)
constexpr int IN_ERR = 96, IN_EPR = 141;                                              // group 3 (2 x 45 values)
constexpr int IN_SIZE = 192;

struct __attribute__((aligned(16))) Smem {
  double RA[NV * LDJ];                  // oMi (FK) -> H -> B columns (rows) and L^-1 g -> T = R^-1 (inequality slots)
  double RB[NV * LDJ];                  // J0 = L^-T, then J = J0 Q ; during assembly (with RC): At, the task stack by DoF
  double RC[PMAX * LDJ];                // Cm: constraint rows (p x 26)
  double in[IN_SIZE];                   // this instance's inputs (q, targets, controller state)
  double pf[WBC_MAX_FRAMES * 3];        // frame origins
  double dv[32], xv[32], npv[32], lv[32], dinv[32], yv[32];
  double cl[64];                        // Cholesky column broadcast; entries 26..63 stay zero
  double bt[48];                        // Cartesian task targets (b of qpb), uniform values
};
constexpr int OFF_OMI = 0;              // RA: oMi[24][12] (dead before H is accumulated)
constexpr int OFF_MC = 24 * 12;         // RA: m*c per joint [32][4]

#define WSYNC() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")

// Diagnostic build (-DWBC_PROFILE): s_memtime stamps at phase boundaries, summed per phase into KernelArgs.prof.
// Never compiled into the shipped library; its run time is not quoted (the stamps serialise the phases).
#ifdef WBC_PROFILE
#define STAMP(ts, i) do { __builtin_amdgcn_sched_barrier(0); (ts)[i] = (unsigned long long)clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(ts, i) do { } while (0)
#endif
enum { T_START = 0, T_FK = 1, T_ASM = 2, T_CHOL = 3, T_INV = 4, T_EQ = 5, T_INEQ = 6, T_END = 7, T_N = 8,
       T_A1 = 8, T_A2 = 9, T_A3 = 10, T_PRE = 11, T_P1 = 12, T_P2 = 13, T_P3 = 14, T_F1 = 15, T_F2 = 16, T_ENTRY = 17, T_NN = 18 };   // sub-stamps inside the task-stack phase (profile build)

// ---------------------------------------------------------------------------------------------- lane helpers
__device__ __forceinline__ double rfl(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rdl(double v, int lane) {  // lane must be wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int rdli(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
// 1 / x and 1 / sqrt(x) from the hardware estimates (~2^-24) + two Newton steps: within an ulp or two of the IEEE sequences at a third
// of their dependent latency (x finite and > 0 — the callers guard). A working-set pass of the dual method waits on five of them.
__device__ __forceinline__ double frcp(double x) { double r = __builtin_amdgcn_rcp(x); r = r * fma(-x, r, 2.0); return r * fma(-x, r, 2.0); }
__device__ __forceinline__ double frsq(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * fma(-0.5 * x * r, r, 1.5);
  return r * fma(-0.5 * x * r, r, 1.5);
}
__device__ __forceinline__ int ctz64(unsigned long long m) { return __ffsll((long long)m) - 1; }

// DPP move of a double (both halves) with a compile-time control word (gfx9 row operations)
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;
// sum / min over lanes 0..31 (the DoF lanes): butterfly inside each 16-lane row, rows 0 and 1 joined by readlane.
// The result is wave-uniform. Lanes 26..31 must carry the neutral element.
__device__ __forceinline__ double wsum(double v) {
  v += dpp<DPP_XOR1>(v);
  v += dpp<DPP_XOR2>(v);
  v += dpp<DPP_HALF_MIRROR>(v);
  v += dpp<DPP_MIRROR>(v);
  return rdl(v, 0) + rdl(v, 16);
}
__device__ __forceinline__ double wmin(double v) {
  v = fmin(v, dpp<DPP_XOR1>(v));
  v = fmin(v, dpp<DPP_XOR2>(v));
  v = fmin(v, dpp<DPP_HALF_MIRROR>(v));
  v = fmin(v, dpp<DPP_MIRROR>(v));
  return fmin(rdl(v, 0), rdl(v, 16));
}

// Scheduling hint: issue the block's LDS reads back to back, then its VALU work. hipcc otherwise serialises
// "ds_read; s_waitcnt; fma" with one or three loads in flight (profiles/r01: 51 % of wave time in s_waitcnt).
#define LDS_THEN_VALU(nread, nvalu) do { __builtin_amdgcn_sched_group_barrier(0x100, nread, 0); \
                                         __builtin_amdgcn_sched_group_barrier(0x002, nvalu, 0); } while (0)

struct double2a { double x, y; } __attribute__((aligned(16)));
__device__ __forceinline__ double2a lds2(const double* p) { return *reinterpret_cast<const double2a*>(p); }
__device__ __forceinline__ void sts2(double* p, double x, double y) { double2a v; v.x = x; v.y = y; *reinterpret_cast<double2a*>(p) = v; }

__device__ __forceinline__ int li_clamp(int lane) { return lane < NV ? lane : NV - 1; }
// model index of instance b: wave-uniform (say so, or every table access becomes a vector load) and CLAMPED to the handle's
// models — a stray value in a caller's device buffer must not turn into an out-of-bounds table read
__device__ __forceinline__ int model_index(const int32_t* model_id, const int b, const int n_models) {
  if (!model_id) return 0;
  const int m = __builtin_amdgcn_readfirstlane(model_id[b]);
  return m < 0 ? 0 : (m >= n_models ? n_models - 1 : m);
}

__device__ __forceinline__ void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

// sin (.x) and cos (.y) for |x| up to a few thousand: Cody–Waite reduction by pi/2 (exact products through FMA) +
// the fdlibm kernel polynomials on [-pi/4, pi/4]; < 1 ulp (ocml's sincos drags in a Payne–Hanek path, ~10x the code).
struct SinCos { double s, c; };
__device__ __forceinline__ SinCos sincos_cw(double x) {
  const double k = rint(x * 0.63661977236758134308);
  double r = fma(-k, 1.5707963267948966, x);
  r = fma(-k, 6.123233995736766e-17, r);
  const double z = r * r;
  const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                    z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
  const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                    z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
  const double s = fma(r * z, ps, r);
  const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
  const int n = ((int)k) & 3;
  SinCos o;
  o.s = (n == 0) ? s : (n == 1) ? c : (n == 2) ? -s : -c;
  o.c = (n == 0) ? c : (n == 1) ? -s : (n == 2) ? -c : s;
  return o;
}

// Eigen::Quaternion::toRotationMatrix without normalisation (what pinocchio's free-flyer uses); q = (x, y, z, w)
__device__ __forceinline__ void quat_to_R(const double* q, double* R) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// Free-flyer part of pin.integrate (Robot_Wrapper4.py:441): M+ = M exp6(v), v = S.xv[0..5] (body twist * dt), current
// placement from the quaternion / xyz staged in S.in; quaternion continuity + first-order renormalisation as in
// pinocchio's SpecialEuclideanOperationTpl<3>::integrate_impl. Uniform arithmetic; lanes 0..6 store.
template <class SM>
__device__ __forceinline__ void integrate_ff(const SM& S, const int lane, double* qn) {
  double R0[9];
  quat_to_R(S.in + IN_Q + 3, R0);
  const double p0[3] = {S.in[IN_Q], S.in[IN_Q + 1], S.in[IN_Q + 2]};
  const double vl[3] = {S.xv[0], S.xv[1], S.xv[2]}, w[3] = {S.xv[3], S.xv[4], S.xv[5]};
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], t = sqrt(t2);
  double a, bq, c;
  if (t < 1e-4) { a = 1 - t2 / 6; bq = 0.5 - t2 / 24; c = 1.0 / 6 - t2 / 120; }
  else { const SinCos sc = sincos_cw(t); a = sc.s / t; bq = (1 - sc.c) / t2; c = (1 - a) / t2; }
  const double wx = w[0], wy = w[1], wz = w[2];
  double Re[9];
  Re[0] = 1 - bq * (wy * wy + wz * wz); Re[1] = -a * wz + bq * wx * wy;       Re[2] = a * wy + bq * wx * wz;
  Re[3] = a * wz + bq * wx * wy;        Re[4] = 1 - bq * (wx * wx + wz * wz); Re[5] = -a * wx + bq * wy * wz;
  Re[6] = -a * wy + bq * wx * wz;       Re[7] = a * wx + bq * wy * wz;        Re[8] = 1 - bq * (wx * wx + wy * wy);
  double wxv[3];
  cross3(w, vl, wxv);
  const double wv = w[0] * vl[0] + w[1] * vl[1] + w[2] * vl[2];
  double pe[3], R1[9], pn[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) pe[i] = a * vl[i] + bq * wxv[i] + c * wv * w[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) R1[3 * i + j] = R0[3 * i] * Re[j] + R0[3 * i + 1] * Re[3 + j] + R0[3 * i + 2] * Re[6 + j];
    pn[i] = p0[i] + (R0[3 * i] * pe[0] + R0[3 * i + 1] * pe[1] + R0[3 * i + 2] * pe[2]);
  }
  double q0, q1, q2, q3;   // Eigen's quaternion-from-matrix
  const double tr = R1[0] + R1[4] + R1[8];
  if (tr > 0) {
    double s = sqrt(tr + 1.0);
    q3 = 0.5 * s; s = 0.5 / s;
    q0 = (R1[7] - R1[5]) * s; q1 = (R1[2] - R1[6]) * s; q2 = (R1[3] - R1[1]) * s;
  } else if (R1[0] >= R1[4] && R1[0] >= R1[8]) {
    double s = sqrt(R1[0] - R1[4] - R1[8] + 1.0);
    q0 = 0.5 * s; s = 0.5 / s;
    q3 = (R1[7] - R1[5]) * s; q1 = (R1[3] + R1[1]) * s; q2 = (R1[6] + R1[2]) * s;
  } else if (R1[4] > R1[0] && R1[4] >= R1[8]) {
    double s = sqrt(R1[4] - R1[8] - R1[0] + 1.0);
    q1 = 0.5 * s; s = 0.5 / s;
    q3 = (R1[2] - R1[6]) * s; q2 = (R1[7] + R1[5]) * s; q0 = (R1[1] + R1[3]) * s;
  } else {
    double s = sqrt(R1[8] - R1[0] - R1[4] + 1.0);
    q2 = 0.5 * s; s = 0.5 / s;
    q3 = (R1[3] - R1[1]) * s; q0 = (R1[2] + R1[6]) * s; q1 = (R1[5] + R1[7]) * s;
  }
  if (q0 * S.in[IN_Q + 3] + q1 * S.in[IN_Q + 4] + q2 * S.in[IN_Q + 5] + q3 * S.in[IN_Q + 6] < 0) { q0 = -q0; q1 = -q1; q2 = -q2; q3 = -q3; }
  const double f = (3 - (q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3)) / 2;
  double outv = 0.0;
  if (lane == 0) outv = pn[0];
  if (lane == 1) outv = pn[1];
  if (lane == 2) outv = pn[2];
  if (lane == 3) outv = q0 * f;
  if (lane == 4) outv = q1 * f;
  if (lane == 5) outv = q2 * f;
  if (lane == 6) outv = q3 * f;
  if (lane < 7) qn[lane] = outv;
}

// ------------------------------------------------------------------------------------------------
// QP: Goldfarb–Idnani dual active set, wavefront form (algebra: tests/gi_variant.py solve_v2).
//   in : H in S.RA (rows 0..25, padded rows = identity), g / lb / ub per lane, Cm in S.RC (p x 26), clb / cub per lane
//   out: x per lane, status, iters.  Replaces qpOASES init/hotstart as called at QP_Wrapper.py:45-48, 70.
// Steps: Cholesky of H in place (RA); J = L^-T into RB (lane c solves L y = e_c); the equalities are absorbed by a
// Householder QR of J'N_e that only updates J; x_eq = J1 y1 - J2 J2'g; then dual active-set iterations for the
// inequalities with T = R^-1 kept (in RA) only for the inequality slots.
// ------------------------------------------------------------------------------------------------
struct QpResult { double x; int status; int iters; int ws_b, ws_r; };   // ws_b / ws_r: final working set, lane = bound / row: 0 inactive, 1 at its lower, 2 at its upper side

// `keep the lowest k set bits of m` (wave-uniform)
__device__ __forceinline__ unsigned long long low_bits(unsigned long long m, int k) {
  while (__popcll(m) > k) m &= ~(1ull << (63 - __clzll((long long)m)));
  return m;
}

// NM = compiled problem-size cap (even, n <= NM <= 26): register-array sizes and loop trip counts; SM = LDS layout
// (Smem or the compact SmemC); CS = row stride of the constraint matrix S.RC
// WARM: compiled with the warm start (the analogue of qpOASES' hotstart, QP_Wrapper.py:55-73): ws_b_in / ws_r_in
// carry the previous tick's final working set in the same per-lane code as QpResult.ws_b / ws_r (algebra: tests/gi_variant.py
// solve_v3). The seeds go through the register-resident Householder QR of the equality block — an order of magnitude cheaper
// per constraint than a dual iteration — but stay droppable (their columns of T = R22^-1 are built along the way); seeds whose
// multiplier comes out negative are removed by the restoration steps in front of the dual iterations.
template <int NM, class SM = Smem, int CS = LDJ, bool WARM = false>
__device__ __forceinline__ QpResult qp_core(SM& S, const double g_in, const double lb_in, const double ub_in,
                                            const double clb_in, const double cub_in, const int n, const int p, const int lane,
                                            unsigned long long* ts, const int dbg_stop = 0, const int ws_b_in = 0, const int ws_r_in = 0) {
  const int li = lane < NM ? lane : NM - 1;
  QpResult res;
  res.status = WBC_QP_OPTIMAL;
  res.iters = 0;
  res.x = 0.0;
  res.ws_b = res.ws_r = 0;
  double g = g_in, lb = lb_in, ub = ub_in, clb = clb_in, cub = cub_in;
  // a NaN bound would silently drop its constraint (every comparison with it is false): refuse the problem instead
  if (__ballot((lane < n && (lb != lb || ub != ub)) || (lane < p && (clb != clb || cub != cub)))) {
    res.status = WBC_QP_NUMERICAL;
    return res;
  }

  // ---- presolve: variables with lb == ub are fixed (the locked gripper / finger DoF, Robot_Wrapper4.py:627-630).
  // Their rows and columns leave H and C (H_kk = 1, g_k = -value reproduces x_k = value), the value's contribution
  // moves into g and the row bounds. Same solution as carrying them as equality constraints, three fewer columns
  // in the equality factorisation. Counted as working-set changes so that `iters` keeps its meaning.
  const bool fixb = (lane < n) && (lb == ub) && (fabs(lb) < QP_INF);
  const unsigned long long fixm = __ballot(fixb);
  const int nfix = __popcll(fixm);
  if (fixm) {
    const double fv = fixb ? lb : 0.0;
    if (__ballot(fv != 0.0)) {              // non-zero fixed values: shift g and the row bounds
      if (lane < 32) S.yv[lane] = fv;
      WSYNC();
      double gs = 0.0, cs = 0.0;
#pragma unroll
      for (int k = 0; k < NM; k += 2) {
        const double2a h2 = lds2(S.RA + li * LDJ + k); const double2a c2 = lds2(S.RC + ((lane < p) ? lane : 0) * CS + k);
        const double2a f2 = lds2(S.yv + k);
        gs = fma(h2.x, f2.x, fma(h2.y, f2.y, gs)); cs = fma(c2.x, f2.x, fma(c2.y, f2.y, cs));
      }
      g += gs;
      if (lane < p) { clb -= cs; cub -= cs; }
      WSYNC();
    }
    unsigned long long m = fixm;
#pragma unroll 1
    while (m) {
      const int k = ctz64(m); m &= m - 1;
      if (lane < NM) S.RA[lane * LDJ + k] = 0.0;
      if (lane < p) S.RC[lane * CS + k] = 0.0;
    }
    if (fixb) {
#pragma unroll
      for (int k = 0; k < NM; k += 2) sts2(S.RA + lane * LDJ + k, 0.0, 0.0);
      S.RA[lane * LDJ + lane] = 1.0;
      g = -fv;
      lb = -1e30; ub = 1e30;                // no longer a constraint
    }
    WSYNC();
  }

  // ---- row `lane` of H into registers (lanes >= 26 shadow row 25; they never write)
  double h[NM];
#pragma unroll
  for (int k = 0; k < NM; k += 2) { const double2a v = lds2(S.RA + li * LDJ + k); h[k] = v.x; h[k + 1] = v.y; }
  WSYNC();

  // ---- Cholesky H = L L', right-looking, ROTATING registers: at step j register r holds column j + r of the row,
  // so the body is the same for every j (a real loop, ~100 instructions) and the row never leaves the VGPRs.
  // Column j is broadcast through S.cl (zero above entry NM - 1); L itself is never stored (the substitutions ride along).
  double pmin = 1.0;
  // Forward substitutions L y = rhs, one right-hand side per lane, ROTATING registers (same trick):
  //   lane c < NM        : e_c        -> y = column c of L^-1 = row c of J0 = L^-T
  //   lane NM + r, r < p : C_r'       -> y = L^-1 C_r'   (column of B = J0' N for constraint row r)
  //   lane NM + p        : g          -> y = L^-1 g
  // Step k needs column k of L — exactly what Cholesky step k broadcasts. The two sweeps are therefore FUSED: one loop,
  // one broadcast per step, L never stored (the separate substitution sweep re-read it from LDS: a quarter of the reduced
  // tick's LDS instructions; +14 % ticks/s on the sim3 kernel, +16 % on the general path).
  if (lane < 32) S.npv[lane] = (lane < n) ? g : 0.0;
  WSYNC();
  double y[NM];
  {
    const int rl = lane - NM;                           // which right-hand side this lane carries
    const double* src = (rl >= 0 && rl < p) ? (S.RC + rl * CS) : S.npv;
    const bool from_lds = (rl >= 0 && rl <= p);
    double sqn = 0.0;
#pragma unroll
    for (int k = 0; k < NM; k += 2) {
      const double2a v = lds2(src + k);
      y[k] = from_lds ? v.x : ((k == lane) ? 1.0 : 0.0);
      y[k + 1] = from_lds ? v.y : ((k + 1 == lane) ? 1.0 : 0.0);
      sqn = fma(v.x, v.x, fma(v.y, v.y, sqn));
    }
    if (rl >= 0 && rl < p) S.yv[rl & 31] = sqn;         // |C_r|^2 (p <= 24 < 32)
  }
  WSYNC();
#pragma unroll 1
  for (int j = 0; j < NM; ++j) {
    // (skipping the FMAs of the padded steps j >= n — their column of L is e_j — and only rotating the registers measured 2 % SLOWER:
    //  the 2 x 15 register moves cost more than the LDS round trip they avoid; same-box A/B, tools/ab_bench.sh)
    const double pj = rdl(h[0], j);
    pmin = (pj > 0.0) ? fmin(pmin, pj) : -1.0;            // (a NaN pivot must fail the test below; fmin would drop it)
    const double rinv = rsqrt(pj);
    const double l = h[0] * rinv;
    if (lane < NM) S.cl[lane] = l;
    WSYNC();
    const double* cj = S.cl + j;
    double cm[NM];
#pragma unroll
    for (int r = 1; r < NM; ++r) cm[r] = cj[r];
#pragma unroll
    for (int r = 1; r < NM; ++r) h[r - 1] = fma(-l, cm[r], h[r]);
    const double yk = y[0] * rinv;
#pragma unroll
    for (int r = 1; r < NM; ++r) y[r - 1] = fma(-cm[r], yk, y[r]);
    y[NM - 1] = yk;
    LDS_THEN_VALU(NM - 1, 2 * NM - 1);
    h[NM - 1] = 0.0;
    WSYNC();
  }
  STAMP(ts, T_CHOL);
  if (!(pmin > 0.0)) { res.status = WBC_QP_NUMERICAL; return res; }
  if (dbg_stop == 6) { res.x = y[0] + h[0]; return res; }      // ablation timing: fused Cholesky / substitution sweep done

  // ---- constraint bookkeeping
  const bool has_b = lane < n, has_r = lane < p;
  const bool eq_b = has_b && (lb == ub) && (fabs(lb) < QP_INF);
  const bool eq_r = has_r && (clb == cub) && (fabs(clb) < QP_INF);
  const unsigned long long eqm_b = __ballot(eq_b), eqm_r = __ballot(eq_r);
  const int nbe = __popcll(eqm_b), ne = nbe + __popcll(eqm_r);
  if (ne > NM) { res.status = WBC_QP_NUMERICAL; return res; }   // more equalities than unknowns
  // warm start: the carried working set's inequalities (a bound presolved above as fixed is infinite by now and drops out)
  unsigned long long sdm_b = 0, sdm_r = 0;
  int nseed = 0;
  if (WARM) {
    sdm_b = __ballot(has_b && !eq_b && ((ws_b_in == 1 && lb > -QP_INF) || (ws_b_in == 2 && ub < QP_INF)));
    sdm_r = __ballot(has_r && !eq_r && ((ws_r_in == 1 && clb > -QP_INF) || (ws_r_in == 2 && cub < QP_INF)));
    const int cap = ((n < NM) ? n : NM) - ne;               // columns the equality QR still has room for
    sdm_b = low_bits(sdm_b, cap);
    sdm_r = low_bits(sdm_r, cap - __popcll(sdm_b));
    nseed = __popcll(sdm_b) + __popcll(sdm_r);
  }
  const int ntot = ne + nseed;

  // lanes < 26: y = row `lane` of J0.  jf2 = |J0|_F^2
  double sq = 0.0;
#pragma unroll
  for (int k = 0; k < NM; ++k) sq = fma(y[k], y[k], sq);
  const double jf2 = wsum(lane < NM ? sq : 0.0);
  const double cn2 = has_r ? S.yv[lane & 31] : 0.0;     // |C_r|^2 for row = lane
  WSYNC();
  // J0 rows -> RB (bound-type equality columns are read from it), B columns and L^-1 g -> RA rows 0..p
  {
    double* dst = (lane < NM) ? (S.RB + lane * LDJ) : ((lane - NM <= p) ? (S.RA + (lane - NM) * LDJ) : nullptr);
    if (dst) {
#pragma unroll
      for (int k = 0; k < NM; k += 2) sts2(dst + k, y[k], y[k + 1]);
    }
  }
  STAMP(ts, T_INV);
  WSYNC();

  // ---- gather row `lane` of B (one register per equality / seed, processing order: equality bounds by index, equality rows,
  // then the seeded bounds and rows, each with the sign of its side) and L^-1 g
  double bq[NM], bg;
  {
    unsigned long long mb = eqm_b, mr = eqm_r, sb = sdm_b, sr = sdm_r;
#pragma unroll
    for (int e = 0; e < NM; ++e) {
      double v = 0.0;
      if (e < ntot) {                                    // uniform
        const double* col;
        double be, n2, sg = 1.0;
        if (mb) { const int c = ctz64(mb); mb &= mb - 1; col = S.RB + c * LDJ; be = rdl(lb, c); n2 = 1.0; }
        else if (mr) { const int c = ctz64(mr); mr &= mr - 1; col = S.RA + c * LDJ; be = rdl(clb, c); n2 = rdl(cn2, c); }
        else if (WARM && sb) {
          const int c = ctz64(sb); sb &= sb - 1; col = S.RB + c * LDJ; n2 = 1.0;
          const bool up = rdli(ws_b_in, c) == 2;
          sg = up ? -1.0 : 1.0; be = up ? -rdl(ub, c) : rdl(lb, c);
        } else {
          const int c = ctz64(sr); sr &= sr - 1; col = S.RA + c * LDJ; n2 = rdl(cn2, c);
          const bool up = rdli(ws_r_in, c) == 2;
          sg = up ? -1.0 : 1.0; be = up ? -rdl(cub, c) : rdl(clb, c);
        }
        v = sg * col[li];
        if (lane == 0) { S.dinv[e] = be; S.lv[e] = n2; }   // (dinv / lv are free after the substitution)
      }
      bq[e] = (lane < n) ? v : 0.0;
    }
    bg = (lane < n) ? S.RA[p * LDJ + li] : 0.0;
  }
  WSYNC();
  // T = 0 in RA (the B columns are in registers now); the seeds' columns of T are written during the QR below
  for (int k = lane; k < NM * LDJ; k += 64) S.RA[k] = 0.0;

  bool act_b = eq_b, act_r = eq_r;        // bound `lane` / row `lane` in the working set (equalities stay in)
  int side_b = 0, side_r = 0;             // side (0 lower, 1 upper) at which bound / row `lane` is active
  double u = 0.0;                         // multiplier of working-set slot `lane` (inequality slots only)
  int a_code = 0;                         // slot `lane`: constraint id | side << 8
  int q = 0, iters = nfix;
  const int max_iter = 10 * (n + p) + 20;
  double* const T = S.RA;

  // ---- equality block: Householder QR of B = J0'N_e with ROTATING columns (bq[0] is always the current column);
  // every reflector is applied at once to the remaining columns, to L^-1 g and to row `lane` of J0 (all in registers).
  // y1 solves R'y1 = b_e incrementally. Nothing but the reflector vector goes through LDS.
  double y1 = 0.0;                        // lane k < q: y1_k
  const bool any_be = nseed > 0 || __ballot((eq_b && lb != 0.0) || (eq_r && clb != 0.0)) != 0;
  int qe = -1;                            // first inequality slot (= number of equalities taken); fixed when the first seed comes up
  unsigned long long sb2 = sdm_b, sr2 = sdm_r, okm_b = 0, okm_r = 0;
#pragma unroll 1
  for (int e = 0; e < ntot; ++e) {
    const bool seed = WARM && e >= ne;    // uniform
    int scode = 0;
    bool take = true;
    if (seed) {
      if (qe < 0) {
        qe = q;
        // Which seeds to take. x0 = the minimiser on the equalities alone (kept in S.npv: the anchor of the refresh further
        // down). A seed is taken only if x0 violates it or comes close to it (within 0.25 max(1, |x0|_inf)): a constraint
        // that is active at the solution almost always is, while a seed far on the feasible side (a velocity bound of tens of
        // rad/s on a joint that hardly moves) would drag the iterate far away — harmless in exact arithmetic, but with
        // cond(H) ~ 1e9 it costs digits along the weakly determined directions (tests/gi_variant.py solve_v3, `far`).
        if (lane < 32) S.dv[lane] = (lane < q) ? y1 : ((lane < n) ? -bg : 0.0);
        WSYNC();
        double x0 = 0.0, x0b = 0.0;
#pragma unroll
        for (int k = 0; k < NM; k += 2) { const double2a v2 = lds2(S.dv + k); x0 = fma(y[k], v2.x, x0); x0b = fma(y[k + 1], v2.y, x0b); }
        x0 += x0b;
        if (lane >= n) x0 = 0.0;
        const double near = 0.25 * fmax(1.0, -wmin(lane < 32 ? -fabs(x0) : 0.0));
        if (lane < 32) S.npv[lane] = x0;
        WSYNC();
        bool okb = false, okr = false;
        if (ws_b_in == 1) okb = (x0 - lb) <= near; else if (ws_b_in == 2) okb = (ub - x0) <= near;
        if (p > 0) {
          double v = 0.0, vb = 0.0;
#pragma unroll
          for (int k = 0; k < NM; k += 2) {
            const double2a c2 = lds2(S.RC + (has_r ? lane : 0) * CS + k); const double2a x2 = lds2(S.npv + k);
            v = fma(c2.x, x2.x, v); vb = fma(c2.y, x2.y, vb);
          }
          v += vb;
          if (ws_r_in == 1) okr = (v - clb) <= near; else if (ws_r_in == 2) okr = (cub - v) <= near;
        }
        okm_b = __ballot(has_b && okb);
        okm_r = __ballot(has_r && okr);
      }
      if (sb2) { const int c = ctz64(sb2); sb2 &= sb2 - 1; scode = c | ((rdli(ws_b_in, c) == 2) ? 256 : 0); take = (okm_b >> c) & 1ull; }
      else { const int c = ctz64(sr2); sr2 &= sr2 - 1; scode = (n + c) | ((rdli(ws_r_in, c) == 2) ? 256 : 0); take = (okm_r >> c) & 1ull; }
    } else ++iters;
    const double d = bq[0];
    const double zn = wsum(lane >= q ? d * d : 0.0);
    const double dy = any_be ? wsum(lane < q ? d * y1 : 0.0) : 0.0;   // y1 stays 0 when every right-hand side is 0
    const double b_e = S.dinv[e], np2 = S.lv[e];
    double beta = 0.0, v = 0.0;
    if (take && zn > 100.0 * n * EPS2 * jf2 * np2) {
      if (seed) ++iters;
      const double dq = rdl(d, q);
      const double sz = sqrt(zn);
      const double delta = (dq >= 0.0) ? -sz : sz;
      const double vv = 2.0 * (zn - delta * dq);
      v = (lane == q) ? d - delta : ((lane > q) ? d : 0.0);     // Householder vector, zero below slot q
      beta = (vv > 0.0) ? 2.0 / vv : 0.0;
      const double yq = (b_e - dy) / delta;
      if (lane == q) y1 = yq;
      if (seed) {
        // the seed stays droppable: column q of T = R22^-1 is (-T r / delta, 1 / delta) with r = the column's entries on the
        // inequality slots [qe, q) — the same append the dual method's add step makes
        if (lane < 32) S.yv[lane] = (lane >= qe && lane < q) ? d : 0.0;
        WSYNC();
        double acc = 0.0;
#pragma unroll 1
        for (int j = qe; j < q; ++j) acc = fma(T[li * LDJ + j], S.yv[j], acc);
        const double idel = 1.0 / delta;
        if (lane >= qe && lane < q) T[lane * LDJ + q] = -acc * idel;
        if (lane == q) { T[lane * LDJ + q] = idel; a_code = scode; }
        const int sc = scode & 255, sd = scode >> 8;
        if (sc >= n) { if (lane == sc - n) { act_r = true; side_r = sd; } } else { if (lane == sc) { act_b = true; side_b = sd; } }
      }
      ++q;
    } else if (!seed && !(fabs(dy - b_e) <= 1e-9 * fmax(1.0, fabs(b_e)))) {   // dependent and inconsistent (a dependent seed is just not taken)
      res.status = WBC_QP_INFEASIBLE; res.iters = iters; return res;
    }
    if (lane < 32) S.dv[lane] = v;
    WSYNC();
    // remaining columns (rotated down by one) and L^-1 g
    const int left = ntot - 1 - e;        // columns still to come
#pragma unroll
    for (int r = 1; r < NM; ++r) {
      if (((r - 1) & 3) == 0 && r > left) break;      // uniform: whole groups of four past the last column are skipped
      const double tau = wsum(v * bq[r]) * beta;
      bq[r - 1] = fma(-tau, v, bq[r]);
    }
    bg = fma(-wsum(v * bg) * beta, v, bg);
    // row `lane` of J0:  row <- row - (row . v) beta v'
    if (beta != 0.0) {
      // (v is zero below the slot it was built for; q was already advanced, so entries k < q - 1 can be skipped
      //  in groups of eight with one uniform branch per group)
      double vk[NM], w = 0.0, w2 = 0.0;
#pragma unroll
      for (int k = 0; k < NM; k += 2) { const double2a v2 = lds2(S.dv + k); vk[k] = v2.x; vk[k + 1] = v2.y; }
      LDS_THEN_VALU(NM / 2, 0);
#pragma unroll
      for (int k0 = 0; k0 < NM; k0 += 8) {
        if (k0 + 8 < q) continue;
#pragma unroll
        for (int k = k0; k < k0 + 8 && k < NM; k += 2) { w = fma(y[k], vk[k], w); w2 = fma(y[k + 1], vk[k + 1], w2); }
      }
      w = (w + w2) * beta;
#pragma unroll
      for (int k0 = 0; k0 < NM; k0 += 8) {
        if (k0 + 8 < q) continue;
#pragma unroll
        for (int k = k0; k < k0 + 8 && k < NM; ++k) y[k] = fma(-w, vk[k], y[k]);
      }
    }
    WSYNC();
  }
  if (qe < 0) qe = q;
  // ---- x_eq = J1 y1 - J2 (J2' g):  bg now holds J'g
  if (lane < 32) S.dv[lane] = (lane < q) ? y1 : ((lane < n) ? -bg : 0.0);
  // J = J0 Q -> RB for the inequality phase
  if (lane < NM) {
#pragma unroll
    for (int k = 0; k < NM; k += 2) sts2(S.RB + lane * LDJ + k, y[k], y[k + 1]);
  }
  if (WARM && q > qe) {   // multipliers of the seeded slots: u = T (y1 + J'g) over [qe, q)
    if (lane < 32) S.yv[lane] = (lane >= qe && lane < q) ? y1 + bg : 0.0;
  }
  WSYNC();
  double x = 0.0, x2s = 0.0;
#pragma unroll
  for (int k = 0; k < NM; k += 2) { const double2a v2 = lds2(S.dv + k); x = fma(y[k], v2.x, x); x2s = fma(y[k + 1], v2.y, x2s); }
  LDS_THEN_VALU(NM / 2, NM);
  x += x2s;
  if (lane >= n) x = 0.0;
  if (WARM && q > qe) {
    double acc = 0.0;
#pragma unroll 1
    for (int j = qe; j < q; ++j) acc = fma(T[li * LDJ + j], S.yv[j], acc);
    if (lane >= qe && lane < q) u = acc;
    WSYNC();
  }
  double* const J = S.RB;
  const double* const Cm = S.RC;
  STAMP(ts, T_EQ);
  if (dbg_stop == 7) { res.x = x; return res; }                // ablation timing: equality phase and x_eq done

  // ---- inequality phase. With seeds taken, RESTORATION first: while a seeded slot's multiplier is negative, the most
  // negative one is dropped and the iterate moved to the minimiser on the remaining set (the add step of the dual method read
  // backwards: x <- x - u_l z, u <- u + u_l r with z, r of the dropped constraint on the new factors); what is left is an S-pair
  // (x minimises on the working set, u >= 0) and the dual iterations start from it.
  bool restoring = WARM && q > qe;
  bool did_restore = false, refreshed = false;
#pragma unroll 1
  for (;;) {
    int wc;
    double s_ip = 0.0, u_l = 0.0;
    int drop_l = -1;
    if (WARM && restoring) {
      const bool slot = lane >= qe && lane < q;
      const double um = wmin((lane < 32 && slot) ? u : 0.0);
      if (!(um < 0.0)) {
        restoring = false;
        if (did_restore && !refreshed) {
          // REFRESH. x and u went through the iterates the wrong seeds put them at, and with cond(H) ~ 1e9 that costs
          // digits; the factors J and T did not (orthogonal updates only). Rebuild x and u from them: with s_j = b_j - n_j'x0
          // the slacks of the remaining slots at the equalities-only minimiser x0 (S.npv),
          //   w = T's,  x = x0 + J[:, qe:q] w,  u = T w      — then one more restoration pass on the accurate multipliers.
          refreshed = true;
          const double x0 = S.npv[lane & 31];
          double v0 = 0.0;
          if (p > 0) {
            double vb = 0.0;
#pragma unroll
            for (int k = 0; k < NM; k += 2) {
              const double2a c2 = lds2(Cm + (has_r ? lane : 0) * CS + k); const double2a x2 = lds2(S.npv + k);
              v0 = fma(c2.x, x2.x, v0); vb = fma(c2.y, x2.y, vb);
            }
            v0 += vb;
          }
          if (lane < 32) {
            S.xv[lane] = has_b ? (side_b ? x0 - ub : lb - x0) : 0.0;
            S.yv[lane] = has_r ? (side_r ? v0 - cub : clb - v0) : 0.0;
          }
          WSYNC();
          const int cc = a_code & 255;
          const double sj = slot ? ((cc < n) ? S.xv[cc & 31] : S.yv[(cc - n) & 31]) : 0.0;
          if (lane < 32) S.dv[lane] = sj;
          WSYNC();
          double w = 0.0;
#pragma unroll 1
          for (int j = qe; j < q; ++j) w = fma(T[j * LDJ + li], S.dv[j], w);
          if (!slot) w = 0.0;
          WSYNC();
          if (lane < 32) S.dv[lane] = w;
          WSYNC();
          double xa = 0.0, ua = 0.0;
#pragma unroll 1
          for (int k = qe; k < q; ++k) { const double wk = S.dv[k]; xa = fma(J[li * LDJ + k], wk, xa); ua = fma(T[li * LDJ + k], wk, ua); }
          x = (lane < n) ? x0 + xa : 0.0;
          u = slot ? ua : 0.0;
          WSYNC();
          restoring = true;
        }
        continue;
      }
      did_restore = true;
      drop_l = ctz64(__ballot(slot && u == um));
      u_l = um;
      wc = rdli(a_code, drop_l);
    } else {
      // most violated inactive inequality
      if (lane < 32) S.xv[lane] = x;
      WSYNC();
      double best = 0.0; int code = -1;
      if (has_b && !act_b && !eq_b) {
        if (lb > -QP_INF) { const double s = x - lb; if (s < -1e-9 * fmax(1.0, fabs(lb)) && s < best) { best = s; code = lane; } }
        if (ub < QP_INF) { const double s = ub - x; if (s < -1e-9 * fmax(1.0, fabs(ub)) && s < best) { best = s; code = lane | 256; } }
      }
      if (p > 0) {
        double v = 0.0, vb = 0.0;
#pragma unroll
        for (int k = 0; k < NM; k += 2) {
          const double2a c2 = lds2(Cm + (has_r ? lane : 0) * CS + k); const double2a x2 = lds2(S.xv + k);
          v = fma(c2.x, x2.x, v); vb = fma(c2.y, x2.y, vb);
        }
        LDS_THEN_VALU(NM, NM);
        v += vb;
        if (has_r && !act_r && !eq_r) {
          if (clb > -QP_INF) { const double s = v - clb; if (s < -1e-9 * fmax(1.0, fabs(clb)) && s < best) { best = s; code = n + lane; } }
          if (cub < QP_INF) { const double s = cub - v; if (s < -1e-9 * fmax(1.0, fabs(cub)) && s < best) { best = s; code = (n + lane) | 256; } }
        }
      }
      const double worst = wmin(lane < 32 ? best : 0.0);
      if (!(worst < 0.0)) break;                          // primal feasible -> optimal
      const int wl = ctz64(__ballot(lane < 32 && best == worst));
      wc = rdli(code, wl);
      s_ip = worst;
    }
    const int ip = wc & 255, ip_side = (wc >> 8) & 1;
    const double b_ip = (ip < n) ? rdl(ip_side ? -ub : lb, ip) : rdl(ip_side ? -cub : clb, ip - n);
    const double sgn = ip_side ? -1.0 : 1.0;
    const bool is_row = ip >= n;
    const int rr = is_row ? ip - n : 0;
    const double np2 = is_row ? rdl(cn2, rr) : 1.0;
    double u_ip = 0.0;

#pragma unroll 1
    for (;;) {
      if (++iters > max_iter) { res.status = WBC_QP_MAX_ITER; goto done; }
      if (drop_l >= 0) {
        // ---- drop slot l: Givens sequence read off the removed row of T, applied to columns of T and J
        const int l = drop_l;
        drop_l = -1;
        const int lc = rdli(a_code, l) & 255;
        if (lc >= n) { if (lane == lc - n) act_r = false; } else { if (lane == lc) act_b = false; }
        if (lane < 32) { S.yv[lane] = u; S.lv[lane] = (double)a_code; }     // shift slots l+1.. down by one (rare path)
        WSYNC();
        if (lane >= l && lane < q - 1) { u = S.yv[lane + 1]; a_code = (int)S.lv[lane + 1]; }
        if (lane == q - 1) { u = 0.0; a_code = 0; }
        const int srow = (li >= l) ? ((li + 1 < NM) ? li + 1 : li) : li;   // old row feeding new row `lane`
        double tx = T[srow * LDJ + l];
        double jx = J[li * LDJ + l];
        double hrun = T[l * LDJ + l];
#pragma unroll 1
        for (int k = l; k < q - 1; ++k) {
          const double tb = T[l * LDJ + k + 1];
          const double nrm2 = fma(hrun, hrun, tb * tb);
          double c_ = 1.0, s_ = 0.0, rho = 0.0;
          if (nrm2 > 0.0) { const double ri = rsqrt(nrm2); c_ = tb * ri; s_ = -hrun * ri; rho = nrm2 * ri; }
          hrun = rho;
          const double ty = T[srow * LDJ + k + 1];
          const double jy = J[li * LDJ + k + 1];
          WSYNC();
          if (lane >= qe && lane < q - 1) T[lane * LDJ + k] = fma(c_, tx, s_ * ty);
          if (lane < n) J[lane * LDJ + k] = fma(c_, jx, s_ * jy);
          tx = fma(-s_, tx, c_ * ty);
          jx = fma(-s_, jx, c_ * jy);
        }
        WSYNC();
        if (lane < q) T[lane * LDJ + q - 1] = 0.0;      // dropped last column, and the vacated last row
        if (lane < q) T[(q - 1) * LDJ + lane] = 0.0;
        if (lane < n) J[lane * LDJ + q - 1] = jx;
        --q;
        WSYNC();
        if (!(WARM && restoring)) {
          const double v = is_row ? wsum(lane < n ? Cm[rr * CS + li] * x : 0.0) : rdl(x, ip);
          s_ip = sgn * v - b_ip;
        }
      }
      double d = 0.0;
      if (is_row) {
        double d2 = 0.0;
#pragma unroll
        for (int i = 0; i < NM; i += 2) {
          const double2a c2 = lds2(Cm + rr * CS + i);
          d = fma(J[i * LDJ + li], c2.x, d);
          d2 = fma(J[(i + 1) * LDJ + li], c2.y, d2);
        }
        LDS_THEN_VALU(NM + NM / 2, NM);
        d = (d + d2) * sgn;
      } else {
        d = sgn * J[ip * LDJ + li];
      }
      if (lane >= n) d = 0.0;
      if (lane < 32) { S.dv[lane] = d; S.yv[lane] = (lane >= q) ? d : 0.0; }
      WSYNC();
      const double zn = wsum((lane >= q && lane < n) ? d * d : 0.0);
      // z = J2 d2 (lane i: row i of J against d restricted to k >= q); r = T d1 (T is zero outside the block of the
      // inequality slots, so the full row product is the product over [qe, q))
      double z = 0.0, zb = 0.0, r = 0.0, rb = 0.0;
#pragma unroll
      for (int k = 0; k < NM; k += 2) {
        const double2a j2 = lds2(J + li * LDJ + k); const double2a y2 = lds2(S.yv + k);
        z = fma(j2.x, y2.x, z); zb = fma(j2.y, y2.y, zb);
      }
      LDS_THEN_VALU(NM, NM);
      z += zb;
      if (q > qe) {
#pragma unroll
        for (int k = 0; k < NM; k += 2) {
          const double2a t2 = lds2(T + li * LDJ + k); const double2a d2 = lds2(S.dv + k);
          r = fma(t2.x, d2.x, r); rb = fma(t2.y, d2.y, rb);
        }
        LDS_THEN_VALU(NM, NM);
        r += rb;
      }
      if (lane < qe || lane >= q) r = 0.0;
      if (lane >= n) z = 0.0;
      if (WARM && restoring) {             // the dropped seed's multiplier u_l < 0 is taken back: minimiser on the reduced set
        x = fma(-u_l, z, x);
        u = fma(u_l, r, u);
        break;
      }
      const bool have_step = zn > 100.0 * n * EPS2 * jf2 * np2;
      const bool cand = (lane >= qe) && (lane < q) && (r > 2.2250738585072014e-308);   // (normal: frcp's estimate of a denormal is inf)
      const double ratio = cand ? u * frcp(r) : INFINITY;
      const double t1 = wmin(lane < 32 ? ratio : INFINITY);
      const int l = (t1 < INFINITY) ? ctz64(__ballot(cand && ratio == t1)) : -1;
      const double t2 = have_step ? -s_ip * frcp(zn) : INFINITY;
      const double t = fmin(t1, t2);
      if (!(t < INFINITY)) { res.status = WBC_QP_INFEASIBLE; goto done; }
      if (have_step) x = fma(t, z, x);
      u = fma(-t, r, u);
      u_ip += t;
      if (have_step && t == t2) {
        // ---- add: Householder P with P d2 = delta e1; J2 <- J2 P; T gets column (-r/delta, 1/delta)
        const double dq = rdl(d, q);
        const double rsz = frsq(zn), sz = zn * rsz;
        const double delta = (dq >= 0.0) ? -sz : sz;
        const double hv = zn - delta * dq;               // v'v / 2
        const double vv = 2.0 * hv;
        if (vv > 0.0) {
          const double w = (z - delta * J[li * LDJ + q]) * frcp(hv);
#pragma unroll
          for (int k = 0; k < NM; k += 2) {
            const double2a j2 = lds2(J + li * LDJ + k); const double2a y2 = lds2(S.yv + k);   // yv = d for k >= q, else 0
            const double v0 = (k == q) ? y2.x - delta : y2.x;
            const double v1 = (k + 1 == q) ? y2.y - delta : y2.y;
            if (lane < n) sts2(J + lane * LDJ + k, fma(-w, v0, j2.x), fma(-w, v1, j2.y));
          }
        }
        const double idel = (dq >= 0.0) ? -rsz : rsz;
        if (lane >= qe && lane < q) T[lane * LDJ + q] = -r * idel;
        if (lane == q) { T[lane * LDJ + q] = idel; u = u_ip; a_code = wc; }
        if (is_row) { if (lane == rr) { act_r = true; side_r = ip_side; } } else { if (lane == ip) { act_b = true; side_b = ip_side; } }
        ++q;
        WSYNC();
        break;
      }
      drop_l = l;                         // blocking slot: dropped at the top of the next pass, then the step is retried
    }
  }
done:
  STAMP(ts, T_INEQ);
  // a QP that was not solved returns x = 0 (the reference's xOpt on its first QP: qpOASES does not write the primal vector
  // of an unsolved problem, QP_Wrapper.py:50, 71-73) — and a roll-out holds still instead of integrating a partial iterate
  if (res.status == WBC_QP_OPTIMAL && __ballot(lane < n && !(fabs(x) <= 1.7976931348623157e308)))
    res.status = WBC_QP_NUMERICAL;                       // NaN / Inf reached the answer (non-finite inputs): never "optimal"
  res.x = (res.status == WBC_QP_OPTIMAL) ? x : 0.0;
  res.iters = iters;
  if (WARM && res.status == WBC_QP_OPTIMAL) {            // the working set the next tick is seeded with (an unsolved QP carries nothing)
    res.ws_b = (act_b && !eq_b) ? 1 + side_b : 0;
    res.ws_r = (act_r && !eq_r) ? 1 + side_r : 0;
  }
  return res;
}

// ------------------------------------------------------------------------------------------------
// J'J on the fp64 matrix cores: H = A'A with v_mfma_f64_16x16x4_f64 (QP_Wrapper.py:17: np.dot(A.T, A)).
// A is m x n (n <= 26, padded to 32 = 2 x 16 columns); k-step s contracts task rows 4s..4s+3.
// Operand maps (cdna_hip_programming.md §3): lane l feeds A_op[i = l&15][k = l>>4] and B_op[k = l>>4][j = l&15], so
// for tile (I, J) both operands are one double per lane: A[4s + (l>>4)][16 I/J + (l&15)]. D: lane l, reg r holds
// D[(l>>4) + 4r][l&15]. Tiles 00, 01, 11 are computed (10 = 01'). `load(r, c)` returns A[r][c] (0 outside);
// column 26 may carry b so that A'b falls out of the same MFMAs (written to S.npv). The tiles land in S.RA = H.
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <class LoadA>
__device__ __forceinline__ void jtj_mfma(Smem& S, const int lane, const int m, LoadA load) {
  v4f64 acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
  const int kq = lane >> 4, c0 = lane & 15;
#pragma unroll 1
  for (int s4 = 0; s4 < m; s4 += 4) {
    const double a0 = load(s4 + kq, c0);
    const double a1 = load(s4 + kq, 16 + c0);
    acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, acc00, 0, 0, 0);
    acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a1, acc01, 0, 0, 0);
    acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, acc11, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = kq + 4 * r, col = c0;
    S.RA[row * LDJ + col] = acc00[r];
    if (16 + col < NV) { S.RA[row * LDJ + 16 + col] = acc01[r]; S.RA[(16 + col) * LDJ + row] = acc01[r]; }
    if (16 + row < NV && 16 + col < NV) S.RA[(16 + row) * LDJ + 16 + col] = acc11[r];
    if (16 + col == NV) { S.npv[row] = acc01[r]; if (16 + row < NV) S.npv[16 + row] = acc11[r]; }
  }
  WSYNC();
}

// H[lane][i] += sum_r At[i][row0 + r] At[lane][row0 + r] for the DoF i in `mask` (the block's support).
template <int NR>
__device__ __forceinline__ void jtj_block(Smem& S, const double* At, const int mtp, const int row0, unsigned mask,
                                          const int lane) {
  const int li = li_clamp(lane);
  double a[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) a[r] = At[li * mtp + row0 + r];
#pragma unroll 1
  while (mask) {                         // two support columns per trip: two independent read-modify-write chains
    const int i0 = __ffs((int)mask) - 1;
    mask &= mask - 1;
    const bool two = mask != 0;
    const int i1 = two ? __ffs((int)mask) - 1 : i0;
    mask &= mask - 1;                    // (0 & anything stays 0)
    double s0 = S.RA[li * LDJ + i0], s1 = S.RA[li * LDJ + i1];
#pragma unroll
    for (int r = 0; r < NR; ++r) { s0 = fma(At[i0 * mtp + row0 + r], a[r], s0); s1 = fma(At[i1 * mtp + row0 + r], a[r], s1); }
    if (lane < NV) { S.RA[lane * LDJ + i0] = s0; if (two) S.RA[lane * LDJ + i1] = s1; }
  }
}

// scipy Rotation.from_matrix(M).as_quat() branch logic (Robot_Wrapper4.py:964-965); M row-major.
// Written out per branch: a dynamically indexed M would be demoted to scratch memory.
__device__ __forceinline__ void R_to_quat(const double* M, double* q) {
  const double tr = M[0] + M[4] + M[8];
  int c = 0;
  double best = M[0];
  if (M[4] > best) { best = M[4]; c = 1; }
  if (M[8] > best) { best = M[8]; c = 2; }
  if (tr > best) c = 3;
  double q0, q1, q2, q3;
  if (c == 3)      { q0 = M[7] - M[5];           q1 = M[2] - M[6];           q2 = M[3] - M[1];           q3 = 1 + tr; }
  else if (c == 0) { q0 = 1 - tr + 2 * M[0];     q1 = M[3] + M[1];           q2 = M[6] + M[2];           q3 = M[7] - M[5]; }
  else if (c == 1) { q1 = 1 - tr + 2 * M[4];     q2 = M[7] + M[5];           q0 = M[1] + M[3];           q3 = M[2] - M[6]; }
  else             { q2 = 1 - tr + 2 * M[8];     q0 = M[2] + M[6];           q1 = M[5] + M[7];           q3 = M[3] - M[1]; }
  const double nn = sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
  q[0] = q0 / nn; q[1] = q1 / nn; q[2] = q2 / nn; q[3] = q3 / nn;
}
__device__ __forceinline__ void quat_mul(const double* a, const double* b, double* r) {
  r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  r[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  r[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}

// ------------------------------------------------------------------------------------------------
// per-lane constants of one model + configuration, kept in registers across the instances a wave processes
// ------------------------------------------------------------------------------------------------
struct LaneConst {
  // joint lane
  int par_off, depth, rev, pris, is_joint, q_idx, a0, a1, a2;
  double t0, t1, t2;
  double mass, c0, c1, c2;
  // frame lane
  int fj_off; double f0, f1, f2;
  // column lane
  int cj_off, col_lin, col_ang, col_q; unsigned subtree;
  // damper (cfg)
  int dq_idx; double d_lo, d_hi, d_vm;
};

__device__ __forceinline__ LaneConst load_lane_const(const DevModel& M, const WbcConfig& cfg, const int lane_true) {
  // Every table entry is fetched unconditionally (clamped index) and selected afterwards: loads behind `cond ? table[i] : 0`
  // were compiled into exec-masked blocks with a full wait between them — three serialised memory round trips at the top
  // of every tick instead of one batch.
  LaneConst c;
  const int ll = lane_true & 31;
  const int jt = M.jtype[ll], par = M.parent[ll], dep = M.depth[ll], iq = M.idx_q[ll];
  const int x0 = M.ax0[ll], x1 = M.ax1[ll], x2 = M.ax2[ll];
  const double t0 = M.tp[ll][0], t1 = M.tp[ll][1], t2 = M.tp[ll][2];
  const double ms = M.mass[ll], c0 = M.com[ll][0], c1 = M.com[ll][1], c2 = M.com[ll][2];
  const int lf = lane_true & 15;
  const int fjn = M.frame_joint[lf];
  const double f0 = M.frame_p[lf][0], f1 = M.frame_p[lf][1], f2 = M.frame_p[lf][2];
  const int cj = M.col_joint[ll], cl_ = M.col_lin[ll], ca = M.col_ang[ll], cq = M.col_q[ll];
  const unsigned st = M.col_subtree[ll];
  const int ld = lane_true < NV ? lane_true : NV - 1;
  const int dq = cfg.damper_qidx[ld];
  const double dlo = cfg.damper_lo[ld], dhi = cfg.damper_hi[ld], dvm = cfg.damper_vmax[ld];
  const int njoints = M.njoints, nv = M.nv;
  c.is_joint = (lane_true >= 2 && lane_true < njoints) ? 1 : 0;
  c.rev = (c.is_joint && jt >= WBC_JT_RX && jt <= WBC_JT_RZ) ? 1 : 0;
  c.pris = (c.is_joint && !c.rev) ? 1 : 0;
  c.par_off = 12 * (c.is_joint ? par : 1);
  c.depth = c.is_joint ? dep : 0;
  c.q_idx = c.is_joint ? iq : 0;
  c.a0 = 3 * x0; c.a1 = 3 * x1; c.a2 = 3 * x2;
  c.t0 = t0; c.t1 = t1; c.t2 = t2;
  c.mass = ms; c.c0 = c0; c.c1 = c1; c.c2 = c2;
  c.fj_off = 12 * fjn; c.f0 = f0; c.f1 = f1; c.f2 = f2;
  c.cj_off = 12 * cj; c.col_lin = cl_; c.col_ang = ca; c.col_q = cq;
  c.subtree = (lane_true < nv) ? st : 0u;
  c.dq_idx = dq; c.d_lo = dlo; c.d_hi = dhi; c.d_vm = dvm;
  return c;
}

// the per-instance inputs, one value per lane per group (coalesced loads), staged into S.in
struct InRegs { double g1, g2, g3a, g3b, pu, qc; };

template <class TI>
__device__ __forceinline__ InRegs load_inputs(const TI& in, const int b, const int lane, const bool has2, const bool has3) {
  InRegs r;
  r.g1 = r.g2 = r.g3a = r.g3b = r.pu = r.qc = 0.0;
  if (in.posture_u && lane < NV) r.pu = in.posture_u[(size_t)b * NV + lane];
  if (in.q_con && lane < NQ) r.qc = in.q_con[(size_t)b * NQ + lane];
  {
    const double* p = nullptr;
    if (lane < 27) p = in.q + (size_t)b * NQ + lane;
    else if (lane >= IN_EET && lane < IN_EET + 15) { if (in.ee_target) p = in.ee_target + (size_t)b * 15 + (lane - IN_EET); }
    else if (lane >= IN_EEP && lane < IN_EEP + 15) { if (in.prev_ee_target) p = in.prev_ee_target + (size_t)b * 15 + (lane - IN_EEP); }
    else if (lane >= IN_BOX && lane < IN_BOX + 4) { if (in.trunk_box_center) p = in.trunk_box_center + (size_t)b * 4 + (lane - IN_BOX); }
    if (p) r.g1 = *p;
  }
  if (has2) {
    const double* p = nullptr;
    const int l2 = lane + 64;
    if (l2 < IN_TP) { if (in.trunk_target) p = in.trunk_target + (size_t)b * 3 + (l2 - IN_TT); }
    else if (l2 < IN_TRE) { if (in.prev_trunk_target) p = in.prev_trunk_target + (size_t)b * 3 + (l2 - IN_TP); }
    else if (l2 < IN_TPR) { if (in.trunk_ref_euler) p = in.trunk_ref_euler + (size_t)b * 3 + (l2 - IN_TRE); }
    else if (l2 < IN_CT) { if (in.trunk_prev_rot) p = in.trunk_prev_rot + (size_t)b * 9 + (l2 - IN_TPR); }
    else if (l2 < IN_CV) { if (in.com_target) p = in.com_target + (size_t)b * 3 + (l2 - IN_CT); }
    else if (l2 < IN_CV + 3) { if (in.com_target_vel) p = in.com_target_vel + (size_t)b * 3 + (l2 - IN_CV); }
    if (p) r.g2 = *p;
  }
  if (has3) {
    if (lane < 45) { r.g3a = in.ee_ref_rot[(size_t)b * 45 + lane]; r.g3b = in.ee_prev_rot[(size_t)b * 45 + lane]; }
  }
  return r;
}
template <class SM>
__device__ __forceinline__ void stage_inputs(SM& S, const InRegs& r, const int lane, const bool has2, const bool has3) {
  S.in[lane] = r.g1;
  if (has2 && lane < 24) S.in[64 + lane] = r.g2;
  if (has3 && lane < 45) { S.in[IN_ERR + lane] = r.g3a; S.in[IN_EPR + lane] = r.g3b; }
}

// ------------------------------------------------------------------------------------------------
// forward kinematics + Jacobian columns of one configuration (updateState's pinocchio calls, Robot_Wrapper4.py:400-405)
// ------------------------------------------------------------------------------------------------
// The handful of model scalars the kinematics read, fetched ONCE per tick by the caller (the sim3 kernel pins them right
// after the model index is known; read where they are used each costs a scalar load + full wait, twice per tick with the
// second FK pass).
struct Hdr { int nq, nv, nj, maxdepth, nframes, trunk_joint; };
__device__ __forceinline__ Hdr load_hdr(const DevModel& M) {
  Hdr h;
  h.nq = M.nq; h.nv = M.nv; h.nj = M.njoints; h.maxdepth = M.maxdepth; h.nframes = M.nframes;
  h.trunk_joint = M.frame_joint[WBC_FR_TRUNK];
  return h;
}
// P1: pin.forwardKinematics. qv = the configuration (LDS), oMi = [joint][12] (R column-major, then p), lane j = joint j.
__device__ __forceinline__ void fk_levels(double* const oMi, const double* const qv, const Hdr& H, const LaneConst& lc,
                                          const int lane) {
  // root free-flyer: R from the quaternion exactly as Eigen's toRotationMatrix, p = xyz
  if (lane == 1) {
    double Rt[9];
    quat_to_R(qv + 3, Rt);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 3; ++r) oMi[12 + 3 * c + r] = Rt[3 * r + c];
    oMi[12 + 9] = qv[0]; oMi[12 + 10] = qv[1]; oMi[12 + 11] = qv[2];
  }
  const double th = lc.is_joint ? qv[lc.q_idx] : 0.0;
  const SinCos sc = sincos_cw(lc.rev ? th : 0.0);
  const double sn = sc.s, cs = sc.c;
  const double pris = lc.pris ? th : 0.0;
  WSYNC();
#pragma unroll 1
  for (int lvl = 2; lvl <= H.maxdepth; ++lvl) {
    if (lc.depth == lvl) {
      const double* Pp = oMi + lc.par_off;
      double Av[3], Bv[3], Cv[3], P[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) { Av[r] = Pp[lc.a0 + r]; Bv[r] = Pp[lc.a1 + r]; Cv[r] = Pp[lc.a2 + r]; P[r] = Pp[9 + r]; }
      double* Po = oMi + 12 * lane;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        Po[lc.a0 + r] = Av[r];
        Po[lc.a1 + r] = cs * Bv[r] + sn * Cv[r];
        Po[lc.a2 + r] = cs * Cv[r] - sn * Bv[r];
        Po[9 + r] = P[r] + Av[r] * (lc.t0 + pris) + Bv[r] * lc.t1 + Cv[r] * lc.t2;
      }
    }
    WSYNC();
  }
}
// P3: column `lane` of data.J (pin.computeJointJacobians, WORLD frame): lin = p_j x axis (revolute) or axis (prismatic)
__device__ __forceinline__ void jac_column(const double* const oMi, const LaneConst& lc, const int lane, const int nv,
                                           double* lin, double* ang) {
  lin[0] = lin[1] = lin[2] = 0.0; ang[0] = ang[1] = ang[2] = 0.0;
  if (lane < nv) {
    const double* Pj = oMi + lc.cj_off;
    const int la = lc.col_lin, aa = lc.col_ang;
    const double pj[3] = {Pj[9], Pj[10], Pj[11]};
    if (aa >= 0) { ang[0] = Pj[3 * aa]; ang[1] = Pj[3 * aa + 1]; ang[2] = Pj[3 * aa + 2]; cross3(pj, ang, lin); }
    if (la >= 0) { lin[0] = Pj[3 * la]; lin[1] = Pj[3 * la + 1]; lin[2] = Pj[3 * la + 2]; }
  }
}
struct FkOut { double lin[3], ang[3], com[3], jc[3], Rtr[9], ptr[3]; };
// P1..P3 + frames + CoM. oMi and (oMi + OFF_MC) are scratch in LDS; frame origins go to S.pf.
template <class SM>
__device__ __forceinline__ void fk_pass(SM& S, double* const oMi, const double* const qv, const Hdr& H,
                                        const LaneConst& lc, const bool need_com, const int lane, FkOut& o,
                                        unsigned long long* ts = nullptr) {
  const int nv = H.nv, nj = H.nj;
  fk_levels(oMi, qv, H, lc, lane);
#ifdef WBC_PROFILE
  if (ts) STAMP(ts, T_F1);
#endif
  // ---- P2: frame origins, pin.updateFramePlacements (Robot_Wrapper4.py:405); frames carry no rotation offset
  if (lane < H.nframes) {
    const double* Pj = oMi + lc.fj_off;
#pragma unroll
    for (int r = 0; r < 3; ++r) S.pf[3 * lane + r] = Pj[9 + r] + Pj[r] * lc.f0 + Pj[3 + r] * lc.f1 + Pj[6 + r] * lc.f2;
  }
  double* const mc = oMi + OFF_MC;
  if (need_com) {   // m_j * c_j (world) per joint, pin.jacobianCenterOfMass's subtree pass (Robot_Wrapper4.py:670)
    if (lane >= 1 && lane < nj) {
      const double* Pj = oMi + 12 * lane;
#pragma unroll
      for (int r = 0; r < 3; ++r) mc[4 * lane + r] = lc.mass * (Pj[9 + r] + Pj[r] * lc.c0 + Pj[3 + r] * lc.c1 + Pj[6 + r] * lc.c2);
      mc[4 * lane + 3] = lc.mass;
    }
  }
  WSYNC();
  jac_column(oMi, lc, lane, nv, o.lin, o.ang);
#ifdef WBC_PROFILE
  if (ts) STAMP(ts, T_F2);
#endif
  o.com[0] = o.com[1] = o.com[2] = 0.0; o.jc[0] = o.jc[1] = o.jc[2] = 0.0;   // whole-body CoM (uniform), column of Jcom
  if (need_com) {
    double ms = 0, s0 = 0, s1 = 0, s2 = 0;
#pragma unroll 1
    for (int j = 1; j < nj; ++j) {
      const double f = ((lc.subtree >> j) & 1u) ? 1.0 : 0.0;
      s0 = fma(f, mc[4 * j], s0); s1 = fma(f, mc[4 * j + 1], s1); s2 = fma(f, mc[4 * j + 2], s2); ms = fma(f, mc[4 * j + 3], ms);
    }
    const double Mt = rdl(ms, 0);
    o.com[0] = rdl(s0, 0) / Mt; o.com[1] = rdl(s1, 0) / Mt; o.com[2] = rdl(s2, 0) / Mt;
    if (lane < nv && ms > 0.0) {
      const double cs_[3] = {s0 / ms, s1 / ms, s2 / ms};
      double wxc[3];
      cross3(o.ang, cs_, wxc);
      const double f = ms / Mt;
#pragma unroll
      for (int r = 0; r < 3; ++r) o.jc[r] = f * (o.lin[r] + wxc[r]);
    }
  }
  // trunk frame (imu): rotation of its supporting joint, uniform read
  const double* Pj = oMi + 12 * H.trunk_joint;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) o.Rtr[3 * r + c] = Pj[3 * c + r];
  o.ptr[0] = S.pf[3 * WBC_FR_TRUNK]; o.ptr[1] = S.pf[3 * WBC_FR_TRUNK + 1]; o.ptr[2] = S.pf[3 * WBC_FR_TRUNK + 2];
}

// ------------------------------------------------------------------------------------------------
// Structural presolve of the contact equalities (the MI355X-side replacement for carrying them through the factorisation).
// A stance foot's rows  Jc_e qdot = 0  (EEConstraint, Robot_Wrapper4.py:757-761) touch the 6 base DoF and that leg's own
// 3 DoF only, so the leg velocities are a linear function of the base velocity:  qdot_leg_e = G_e qdot_base,
// G_e = -K_e^-1 B_e  with K_e the 3 x 3 leg block and B_e the 3 x 6 base block of Jc_e. Substituting x = Z y
// (y = base + every DoF that is not an eliminated leg) gives an equivalent QP in n - 3 f unknowns with NO contact
// equalities:  H' = Z'HZ, g' = Z'g, remaining rows C' = CZ, and the eliminated legs' velocity bounds become f x 3 general
// rows  lb_leg <= G_e y_base <= ub_leg.  For A1 + wx200 with four stance feet: 26 unknowns / 12 equalities -> 14 / 0, and
// the dense phases (Cholesky, L^-1, equality QR) shrink accordingly (qp_core<16>).
// It is applied only where it costs no accuracy: when no active task touches the eliminated legs (the sim3 tick: Grip
// task + posture; DevPlan.enabled, decided by wbc_batch_configure) H_ll = d^2 I and H_lf = 0 exactly, so
// H' = H_ff + d^2 G'G on the base block — no cancellation. With foot / CoM tasks on, Z'HZ is 10^3 x worse conditioned than H
// (|G| ~ 100 in the WORLD-frame rows) and the general path is kept. Falls back (returns false) at run time when a leg
// block is numerically singular. Same minimiser as the full problem (tests compare both paths against the oracle).
// LDS: G lives at RB[16 LDJ ..] (rows >= 16 of RB are never touched by qp_core<16>, so it survives the solve).
// ------------------------------------------------------------------------------------------------
constexpr int NR = WBC_PLAN_NR;        // compiled size cap of the reduced problem (16)
constexpr int GS = 10;                 // row stride of G: 6 base columns + up to 4 extra unknowns (one per rank-deficient stance-leg block)
// WARM: the carried working set (ws0 / ws1, FULL-problem indexing: KernelArgs.ws_in) is mapped into the reduced problem — reduced
// variable k is DoF Fd[k], reduced row r is the r-th kept row or, from p_keep on, the velocity bound of eliminated leg DoF legd[r - p_keep]
// — and the final one mapped back, so that res.ws_b / res.ws_r come out in the caller's indexing (lane = DoF / original constraint row),
// like process_sim3's.
template <bool WARM = false, class KA = KernelArgs>
__device__ __forceinline__ bool contact_presolve(Smem& S, const KA& A, const DevModel& M, const WbcConfig& cfg,
                                                 const DevPlan& P, const double dpost, const double g, const double lb,
                                                 const double ub, const double clb, const double cub, const int lane,
                                                 unsigned long long* ts, QpResult& res, const unsigned long long ws0 = 0ull,
                                                 const unsigned long long ws1 = 0ull) {
  if (!A.presolve || !P.enabled) return false;
  const int nv = M.nv, p = A.prows;
  const int nelim = P.nelim, n_red = P.n_red, nl = 3 * nelim;
  double* const Gm = S.RB + NR * LDJ;            // [12][GS]: row l = eliminated leg DoF l, columns = base DoF (the extra columns stay unused here)
  double* const Cm = S.RC;
  // the plan's index maps, fetched up front in one batch of scalar loads (loaded where they are used, each value costs
  // its own s_load + full wait inside the dependent chain: profiles/r01_phase_cycles_v9a.json, 29k cycles of presolve)
  int legd[12], Fd[NR], rowstart[4];
#pragma unroll
  for (int l = 0; l < 12; ++l) legd[l] = P.legd[l];
#pragma unroll
  for (int k = 0; k < NR; ++k) Fd[k] = P.Fd[k];
#pragma unroll
  for (int f = 0; f < 4; ++f) rowstart[f] = P.rowstart[f];
  const unsigned elimrows = P.elimrows, legrows = P.legrows;
#pragma unroll
  for (int l = 0; l < 12; ++l) asm volatile("" : "+s"(legd[l]));
#pragma unroll
  for (int k = 0; k < NR; ++k) asm volatile("" : "+s"(Fd[k]));
#pragma unroll
  for (int f = 0; f < 4; ++f) asm volatile("" : "+s"(rowstart[f]));
  // per-lane views of the maps by select chains over the SGPR copies (a per-lane global load of the plan stalled the
  // wave for thousands of cycles): fj = DoF of reduced variable `lane`; for lane = DoF d: its reduced position or its
  // eliminated-leg index; my_legd = leg DoF whose bound row is row `lane` of the reduced constraint matrix
  const int p_keep = P.p_keep;
  int fj = 0, my_pos = -1, my_l = -1, my_legd = 0;
#pragma unroll
  for (int k = 0; k < NR; ++k) { fj = (lane == k) ? Fd[k] : fj; my_pos = (lane == Fd[k] && k < n_red) ? k : my_pos; }
#pragma unroll
  for (int l = 0; l < 12; ++l) { my_l = (lane == legd[l] && l < nl) ? l : my_l; my_legd = (lane - p_keep == l) ? legd[l] : my_legd; }
  if (lane < 32) { S.npv[lane] = (lane < nv) ? g : 0.0; S.xv[lane] = lb; S.yv[lane] = ub; }

  // ---- G_e = -K_e^-1 B_e, all feet at once: lane 6 f + c owns column c of foot f (K_f^-1 by the adjugate, computed by
  // each of the foot's six lanes)
  bool singular = false;
  {
    const int f = (lane < 24) ? lane / 6 : 0, c = (lane < 24) ? lane - 6 * f : 0;
    int d0 = legd[0], d1 = legd[1], d2 = legd[2], rs = rowstart[0];
#pragma unroll
    for (int t = 1; t < 4; ++t) { const bool m = f == t; d0 = m ? legd[3 * t] : d0; d1 = m ? legd[3 * t + 1] : d1; d2 = m ? legd[3 * t + 2] : d2; rs = m ? rowstart[t] : rs; }
    const double* r0 = Cm + rs * LDJ; const double* r1 = r0 + LDJ; const double* r2 = r1 + LDJ;
    const double k00 = r0[d0], k01 = r0[d1], k02 = r0[d2], k10 = r1[d0], k11 = r1[d1], k12 = r1[d2],
                 k20 = r2[d0], k21 = r2[d1], k22 = r2[d2];
    const double b0 = r0[c], b1 = r1[c], b2 = r2[c];
    const double a00 = k11 * k22 - k12 * k21, a01 = k02 * k21 - k01 * k22, a02 = k01 * k12 - k02 * k11;
    const double a10 = k12 * k20 - k10 * k22, a11 = k00 * k22 - k02 * k20, a12 = k02 * k10 - k00 * k12;
    const double a20 = k10 * k21 - k11 * k20, a21 = k01 * k20 - k00 * k21, a22 = k00 * k11 - k01 * k10;
    const double det = k00 * a00 + k01 * a10 + k02 * a20;
    const double sc = fabs(k00) + fabs(k01) + fabs(k02) + fabs(k10) + fabs(k11) + fabs(k12) + fabs(k20) + fabs(k21) + fabs(k22);
    const bool live = lane < 6 * nelim;
    singular = __ballot(live && !(fabs(det) > A.sing_tol * sc * sc * sc)) != 0;   // a leg block (nearly) rank deficient: general path
    const double id = -1.0 / det;
    if (lane < 24) {
      Gm[(3 * f + 0) * GS + c] = live ? id * (a00 * b0 + a01 * b1 + a02 * b2) : 0.0;
      Gm[(3 * f + 1) * GS + c] = live ? id * (a10 * b0 + a11 * b1 + a12 * b2) : 0.0;
      Gm[(3 * f + 2) * GS + c] = live ? id * (a20 * b0 + a21 * b1 + a22 * b2) : 0.0;
    }
  }
  if (singular) return false;
  WSYNC();
  STAMP(ts, T_P1);
  // per-lane column of G (lanes >= 6: zero), kept for H', g' and C'
  double gcol[12];
#pragma unroll
  for (int l = 0; l < 12; ++l) gcol[l] = (lane < 6) ? Gm[l * GS + lane] : 0.0;   // rows >= nl are zero

  // g' = Z'g
  double g_red = S.npv[fj];
#pragma unroll
  for (int l = 0; l < 12; ++l) g_red = fma(gcol[l], S.npv[legd[l]], g_red);
  if (lane >= n_red) g_red = 0.0;
  STAMP(ts, T_P2);
  // ---- C' = C Z for the rows that stay (in their order), then the eliminated legs' bounds as rows G_l
  double nclb = 0.0, ncub = 0.0;
  int i2 = 0;
#pragma unroll 1
  for (int i = 0; i < p; ++i) {
    if ((elimrows >> i) & 1u) continue;
    double v = (lane < n_red) ? Cm[i * LDJ + fj] : 0.0;
    if ((legrows >> i) & 1u) {                          // rows without leg support (the trunk box) need no G
#pragma unroll
      for (int l = 0; l < 12; ++l) v = fma(gcol[l], Cm[i * LDJ + legd[l]], v);
    }
    const double bl = rdl(clb, i), bu = rdl(cub, i);
    WSYNC();
    if (lane < NV) Cm[i2 * LDJ + lane] = v;
    if (lane == i2) { nclb = bl; ncub = bu; }
    WSYNC();
    ++i2;
  }
  if (cfg.use_bounds) {
#pragma unroll
    for (int l = 0; l < 12; ++l) {
      if (l < nl) { if (lane < NV) Cm[(i2 + l) * LDJ + lane] = gcol[l]; }
    }
    if (lane >= i2 && lane < i2 + nl) { nclb = S.xv[my_legd]; ncub = S.yv[my_legd]; }
    i2 += nl;
  }
  const double lb_red = (lane < n_red) ? S.xv[fj] : 0.0, ub_red = (lane < n_red) ? S.yv[fj] : 0.0;
  WSYNC();
  STAMP(ts, T_P3);
  // ---- row `lane` of H' (lanes < n_red), identity padding up to NR:  H_ff  +  d^2 G'G on the base block
  double hr[NR];
  {
    double gg[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int l = 0; l < 12; ++l) {
#pragma unroll
      for (int c = 0; c < 6; ++c) gg[c] = fma(gcol[l], Gm[l * GS + c], gg[c]);
    }
    const double d2 = dpost * dpost;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      double v = S.RA[fj * LDJ + Fd[k]];
      if (k < 6) v = fma(d2, gg[k], v);
      hr[k] = (lane < n_red && k < n_red) ? v : ((k == lane) ? 1.0 : 0.0);
    }
  }
  WSYNC();
  // ---- H' into RA (rows and columns < NR are all qp_core<NR> reads; the rest is cleared so that nothing of the 26-wide
  // H survives next to it — and, as a side effect, this store burst keeps hipcc's register allocation of the general
  // kernel at 200 VGPRs: without it the same code spills 20)
  for (int k = lane; k < NV * LDJ; k += 64) S.RA[k] = 0.0;
  WSYNC();
  if (lane < NR) {
#pragma unroll
    for (int k = 0; k < NR; k += 2) sts2(S.RA + lane * LDJ + k, hr[k], hr[k + 1]);
  }
  WSYNC();
  STAMP(ts, T_PRE);
  int sd_b = 0, sd_r = 0;
  if (WARM) {
    int my_orig = -1, cnt = 0;                        // original index of kept row `lane`
#pragma unroll 1
    for (int i = 0; i < p; ++i) { if (!((elimrows >> i) & 1u)) { my_orig = (cnt == lane) ? i : my_orig; ++cnt; } }
    if (lane < n_red) sd_b = (int)(((ws0 >> fj) & 1ull) | (((ws0 >> (32 + fj)) & 1ull) << 1));
    if (my_orig >= 0) sd_r = (int)(((ws1 >> my_orig) & 1ull) | (((ws1 >> (32 + my_orig)) & 1ull) << 1));
    else if (cfg.use_bounds && lane >= p_keep && lane < p_keep + nl) sd_r = (int)(((ws0 >> my_legd) & 1ull) | (((ws0 >> (32 + my_legd)) & 1ull) << 1));
    if (sd_b == 3) sd_b = 0;
    if (sd_r == 3) sd_r = 0;
  }
  if (n_red <= 12) res = qp_core<12, Smem, LDJ, WARM>(S, g_red, lb_red, ub_red, nclb, ncub, n_red, i2, lane, ts, 0, sd_b, sd_r);   // (qp_core's sweeps cost ~NM^2)
  else res = qp_core<NR, Smem, LDJ, WARM>(S, g_red, lb_red, ub_red, nclb, ncub, n_red, i2, lane, ts, 0, sd_b, sd_r);
  res.iters += nl + P.nlock;                         // the eliminated equalities and the locked DoF, so that `iters` keeps its meaning
  // ---- x = Z y
  WSYNC();
  if (lane < 32) { S.xv[lane] = (lane < n_red) ? res.x : 0.0; if (WARM) { S.lv[lane] = (double)res.ws_b; S.dinv[lane] = (double)res.ws_r; } }
  WSYNC();
  if (WARM) {   // the final working set back in full-problem indexing: lane d = DoF d, lane i = original constraint row i
    int cb = 0, cr = 0;
    if (my_pos >= 0) cb = (int)S.lv[my_pos];
    else if (my_l >= 0 && cfg.use_bounds) cb = (int)S.dinv[(p_keep + my_l) & 31];
    if (lane < p && !((elimrows >> lane) & 1u)) cr = (int)S.dinv[__popc(~elimrows & ((1u << lane) - 1u)) & 31];
    res.ws_b = cb; res.ws_r = cr;
  }
  double x = 0.0;
  if (my_pos >= 0) x = S.xv[my_pos];
  else if (my_l >= 0) {
#pragma unroll
    for (int c = 0; c < 6; ++c) x = fma(Gm[my_l * GS + c], S.xv[c], x);
  }
  res.x = (lane < nv) ? x : 0.0;
  return true;
}

// ------------------------------------------------------------------------------------------------
// The contact presolve for configurations whose tasks DO touch the stance legs (foot / trunk / CoM tasks: DevPlan.orth; BASELINE
// configs[1] is one): the explicit G = -K^-1 B of contact_presolve makes Z'HZ up to 10^8 x worse conditioned than H there, so the
// contact equalities E [qd_base; qd_legs] = 0 (E = [B K], 3 rows per stance foot) are eliminated through an ORTHONORMAL basis of
// their null space instead: Householder QR of E' (18 x 12), Z = the last six columns of Q, [qd_base; qd_legs] = Z y~. Then
// cond(Z'HZ) <= cond(H), a rank-deficient K is no special case (E keeps full row rank through B), and the reduced problem has
// n' = 6 + (free DoF outside base and stance legs) unknowns:
//     H' = Z'HZ  (6 x 6 block and 6 x rest strip recomputed, the rest of H kept),   g' = Z'g,
//     kept rows C' = C Z;  the velocity bounds of the base and stance-leg DoF become the 6 + 3 nelim rows of Z (two-sided),
//     the other DoF keep their simple bounds;  qd = Z y.
// Same minimiser as the full problem (tests compare against the oracle's full solve). Returns false (general path) only when two
// contact rows are numerically dependent.  LDS: T = H(:, bl) Z at RB[32..188), Z'T at RB[188..224),
// Z at RB[16 LDJ ..)
// (rows >= 16 of RB are never touched by qp_core<16>).
// ------------------------------------------------------------------------------------------------
// Householder QR of E' (the stance feet's contact rows over [base; stance legs]) -> Z, an orthonormal basis of their null space, one row
// per lane 16 .. 33 written to Zm [18][6] (rows: base DoF 0..5, then eliminated leg DoF l). `rows` + rs[f] * LDJ is the first of
// foot f's three rows (26-wide, as the constraint stage writes them). Returns false when two rows are numerically dependent.
__device__ __forceinline__ bool orth_qr_z(const double* const rows, const int (&rowstart)[4], const int (&legd)[12], const int nelim,
                                          const int nl, const int lane, const double sing_tol, double* const Zm) {
  constexpr int NB = 18;
  const double* const Cm = rows;
  // ---- Householder QR of E' with the coordinates ordered [leg 0, leg 1, .., base]: the reflector of column k = 3 f + r then has
  // support on leg f's coordinates r..2 and the base only, and what it leaves in the other legs' coordinates of a later column
  // is part of R (never read again). So every vector is carried as base[6] + cur[3] (its entries at the current leg's
  // coordinates) whatever its length: lane j < nl = column j of E' (contact row j), lanes 16..33 = the unit vectors, which end up
  // as the rows of Q — their base part is the row of Z (u_i = Q'e_i, Z = Q[:, nl..nl+5]).
  double base[6], kown[3];
  int own_f;
  {
    const int f = (lane < 12) ? lane / 3 : 0, rr = (lane < 12) ? lane - 3 * f : 0;
    int rs = rowstart[0], d0 = legd[0], d1 = legd[1], d2 = legd[2];
#pragma unroll
    for (int t = 1; t < 4; ++t) { const bool m = f == t; rs = m ? rowstart[t] : rs; d0 = m ? legd[3 * t] : d0; d1 = m ? legd[3 * t + 1] : d1; d2 = m ? legd[3 * t + 2] : d2; }
    const double* row = Cm + (rs + rr) * LDJ;
    const bool col = lane < nl;
    const int ui = lane - 16;                    // unit vector index (0..5 base, 6 + l leg coordinate l)
#pragma unroll
    for (int i = 0; i < 6; ++i) base[i] = col ? row[i] : ((ui == i) ? 1.0 : 0.0);
    kown[0] = col ? row[d0] : 0.0; kown[1] = col ? row[d1] : 0.0; kown[2] = col ? row[d2] : 0.0;
    own_f = col ? f : -1;
    if (ui >= 6 && ui < NB) {
      const int l = ui - 6, lf = l / 3, lt = l - 3 * lf;
      own_f = lf; kown[0] = (lt == 0) ? 1.0 : 0.0; kown[1] = (lt == 1) ? 1.0 : 0.0; kown[2] = (lt == 2) ? 1.0 : 0.0;
    }
  }
  double c0 = fma(kown[0], kown[0], fma(kown[1], kown[1], kown[2] * kown[2]));
#pragma unroll
  for (int i = 0; i < 6; ++i) c0 = fma(base[i], base[i], c0);
  bool dependent = false;
#pragma unroll 1
  for (int f = 0; f < nelim; ++f) {
    double cur[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) cur[t] = (own_f == f) ? kown[t] : 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      // every lane forms "its" reflector (branch-free); the pivot lane's is the one that counts and is read out of its registers
      // with v_readlane (the pivot lane index is wave-uniform) — no LDS round trip inside the 12-step chain (measured against the
      // LDS broadcast: 2 % of the C2 tick)
      const int pl = 3 * f + r;
      double sa = base[0] * base[0], sb_ = base[1] * base[1], sc_ = base[2] * base[2];
      sa = fma(base[3], base[3], sa); sb_ = fma(base[4], base[4], sb_); sc_ = fma(base[5], base[5], sc_);
      if (r <= 0) sa = fma(cur[0], cur[0], sa);
      if (r <= 1) sb_ = fma(cur[1], cur[1], sb_);
      sc_ = fma(cur[2], cur[2], sc_);
      const double sig = (sa + sb_) + sc_;       // |x|^2 of the column from its pivot entry down
      const double ek = cur[r];
      // v_rsq_f64 / v_rcp_f64 are good to ~2^-24: one Newton step each leaves beta within ~1e-14 of 2 / v'v and Q orthogonal to
      // that (the IEEE sqrt and division sequences are four times as long)
      double rs = __builtin_amdgcn_rsq(sig);
      rs = rs * fma(-0.5 * sig * rs, rs, 1.5);
      const double nrm = (sig > 0.0) ? sig * rs : 0.0;
      const double alpha = (ek > 0.0) ? -nrm : nrm;
      const double den = fma(-alpha, ek, sig);   // v'v / 2
      double rd = __builtin_amdgcn_rcp(den);
      rd = rd * fma(-den, rd, 2.0);
      if (lane == pl) dependent = dependent || !(sig > sing_tol * sing_tol * c0);
      double vb[6], vl[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int i = 0; i < 6; ++i) vb[i] = rdl(base[i], pl);
      vl[r] = rdl(ek - alpha, pl);
#pragma unroll
      for (int t = r + 1; t < 3; ++t) vl[t] = rdl(cur[t], pl);
      const double beta = rdl((den > 0.0) ? rd : 0.0, pl);
      double wa = vb[0] * base[0], wb = vb[1] * base[1], wc = vb[2] * base[2];
      wa = fma(vb[3], base[3], wa); wb = fma(vb[4], base[4], wb); wc = fma(vb[5], base[5], wc);
      if (r <= 0) wa = fma(vl[0], cur[0], wa);
      if (r <= 1) wb = fma(vl[1], cur[1], wb);
      wc = fma(vl[2], cur[2], wc);
      const double w = ((wa + wb) + wc) * beta;
#pragma unroll
      for (int i = 0; i < 6; ++i) base[i] = fma(-w, vb[i], base[i]);
#pragma unroll
      for (int t = r; t < 3; ++t) cur[t] = fma(-w, vl[t], cur[t]);
    }
  }
  if (__ballot(dependent)) return false;
  if (lane >= 16 && lane < 16 + NB) {
#pragma unroll
    for (int c = 0; c < 6; c += 2) sts2(Zm + (lane - 16) * 6 + c, base[c], base[c + 1]);
  }
  WSYNC();
  return true;
}

// The same null-space basis without the 12-step QR, for the (usual) stance whose leg blocks are well conditioned: G = -K^-1 B by the
// adjugate (all feet at once), then Z = [I; G] S with S = L^-T, L L' = I + G'G — the columns of [I; G] orthonormalised through a 6 x 6
// Cholesky factor instead of through the 18 x 12 Householder sweep (~250 wave-instructions instead of ~1000). What went wrong with the
// explicit G in round 1 was the missing S (cond(Z'HZ) ~ |G|^2 cond(H)), not G itself: with |G| up to 10^4 the orthogonality error is
// 4e-9 and the reduced solve is as accurate as the QR's (tests; tools note in DESIGN.md §3.9). Any leg block with
// |det K| <= 1e-6 (sum |K_ij|)^3 sends the instance to orth_qr_z, for which a rank-deficient K is no special case.
__device__ __forceinline__ bool orth_null_basis(const double* const rows, const int (&rowstart)[4], const int (&legd)[12], const int nelim,
                                                const int nl, const int lane, const double sing_tol, double* const Zm, const bool force_qr) {
  // ---- G (rows 6 + l of Zm for now)
  bool flagged;
  {
    const int f = (lane < 24) ? lane / 6 : 0, c = (lane < 24) ? lane - 6 * f : 0;
    int d0 = legd[0], d1 = legd[1], d2 = legd[2], rs = rowstart[0];
#pragma unroll
    for (int t = 1; t < 4; ++t) { const bool m = f == t; d0 = m ? legd[3 * t] : d0; d1 = m ? legd[3 * t + 1] : d1; d2 = m ? legd[3 * t + 2] : d2; rs = m ? rowstart[t] : rs; }
    const double* r0 = rows + rs * LDJ; const double* r1 = r0 + LDJ; const double* r2 = r1 + LDJ;
    const double k00 = r0[d0], k01 = r0[d1], k02 = r0[d2], k10 = r1[d0], k11 = r1[d1], k12 = r1[d2],
                 k20 = r2[d0], k21 = r2[d1], k22 = r2[d2];
    const double b0 = r0[c], b1 = r1[c], b2 = r2[c];
    const double a00 = k11 * k22 - k12 * k21, a01 = k02 * k21 - k01 * k22, a02 = k01 * k12 - k02 * k11;
    const double a10 = k12 * k20 - k10 * k22, a11 = k00 * k22 - k02 * k20, a12 = k02 * k10 - k00 * k12;
    const double a20 = k10 * k21 - k11 * k20, a21 = k01 * k20 - k00 * k21, a22 = k00 * k11 - k01 * k10;
    const double det = k00 * a00 + k01 * a10 + k02 * a20;
    const double sc = fabs(k00) + fabs(k01) + fabs(k02) + fabs(k10) + fabs(k11) + fabs(k12) + fabs(k20) + fabs(k21) + fabs(k22);
    const bool live = lane < 6 * nelim;
    flagged = force_qr || __ballot(live && !(fabs(det) > fmax(1e-6, sing_tol) * sc * sc * sc)) != 0;
    const double id = -1.0 / det;
    if (lane < 24) {
      Zm[(6 + 3 * f + 0) * 6 + c] = live ? id * (a00 * b0 + a01 * b1 + a02 * b2) : 0.0;
      Zm[(6 + 3 * f + 1) * 6 + c] = live ? id * (a10 * b0 + a11 * b1 + a12 * b2) : 0.0;
      Zm[(6 + 3 * f + 2) * 6 + c] = live ? id * (a20 * b0 + a21 * b1 + a22 * b2) : 0.0;
    }
  }
  if (flagged) { WSYNC(); return orth_qr_z(rows, rowstart, legd, nelim, nl, lane, sing_tol, Zm); }
  WSYNC();
  // ---- M = I + G'G, one entry per lane (rows 0..5 of Zm for now)
  if (lane < 36) {
    const int c = lane / 6, k = lane - 6 * c;
    double m0 = (c == k) ? 1.0 : 0.0, m1 = 0.0;
#pragma unroll
    for (int l = 0; l < 12; l += 2) {
      m0 = fma(Zm[(6 + l) * 6 + c], Zm[(6 + l) * 6 + k], m0);
      m1 = fma(Zm[(7 + l) * 6 + c], Zm[(7 + l) * 6 + k], m1);
    }
    Zm[lane] = m0 + m1;
  }
  WSYNC();
  // ---- L L' = M and Li = L^-1 in registers, the same on every lane (M >= I: no pivot can fail)
  double Lm[6][6], Li[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double2a a = lds2(Zm + 6 * i), b = lds2(Zm + 6 * i + 2), c = lds2(Zm + 6 * i + 4);
    Lm[i][0] = a.x; Lm[i][1] = a.y; Lm[i][2] = b.x; Lm[i][3] = b.y; Lm[i][4] = c.x; Lm[i][5] = c.y;
  }
  double dinv[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double v = Lm[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) v = fma(-Lm[j][k], Lm[j][k], v);
    double rs = __builtin_amdgcn_rsq(v);
    rs = rs * fma(-0.5 * v * rs, rs, 1.5); rs = rs * fma(-0.5 * v * rs, rs, 1.5);
    dinv[j] = rs;
    Lm[j][j] = v * rs;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double w = Lm[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) w = fma(-Lm[i][k], Lm[j][k], w);
      Lm[i][j] = w * rs;
    }
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {             // column c of L^-1: L x = e_c
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (i < c) Li[i][c] = 0.0;
      else {
        double w = (i == c) ? 1.0 : 0.0;
#pragma unroll
        for (int k = c; k < i; ++k) w = fma(-Lm[i][k], Li[k][c], w);
        Li[i][c] = w * dinv[i];
      }
    }
  }
  WSYNC();                                   // every lane has M: rows 0..5 become S = Li'
  // ---- Z = [S; G S], S[c][k] = Li[k][c]
  if (lane < 6) {
    double srow[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      double v = 0.0;
#pragma unroll
      for (int c = 0; c < 6; ++c) v = (lane == c) ? Li[k][c] : v;
      srow[k] = v;
    }
    sts2(Zm + lane * 6, srow[0], srow[1]); sts2(Zm + lane * 6 + 2, srow[2], srow[3]); sts2(Zm + lane * 6 + 4, srow[4], srow[5]);
  }
  if (lane >= 16 && lane < 28) {
    double* zr = Zm + (6 + lane - 16) * 6;
    const double2a a = lds2(zr), b = lds2(zr + 2), c = lds2(zr + 4);
    const double gl[6] = {a.x, a.y, b.x, b.y, c.x, c.y};
    double o[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      double v = 0.0;
#pragma unroll
      for (int c2 = 0; c2 <= k; ++c2) v = fma(gl[c2], Li[k][c2], v);
      o[k] = v;
    }
    sts2(zr, o[0], o[1]); sts2(zr + 2, o[2], o[3]); sts2(zr + 4, o[4], o[5]);
  }
  WSYNC();
  return true;
}

#ifdef ORTH_CUT   // timing cuts (variant builds only: make variant VFLAGS=-DORTH_CUT=k): the presolve returns after stage k with garbage
#define OCUT(k, val) do { if (ORTH_CUT == (k)) { res.x = (val); res.status = 0; res.iters = 0; res.ws_b = res.ws_r = 0; return true; } } while (0)
#else
#define OCUT(k, val) do { } while (0)
#endif
template <class KA>
__device__ __forceinline__ bool contact_presolve_orth(Smem& S, const KA& A, const DevModel& M, const WbcConfig& cfg,
                                                      const DevPlan& P, const double g, const double lb, const double ub,
                                                      const double clb, const double cub, const int lane,
                                                      unsigned long long* ts, QpResult& res, const bool have_h = false) {
  // have_h: orth_direct_assemble has been there: Z and H' are in place (no 26-wide H exists)
  if (!have_h && (!A.presolve || !A.presolve_orth || !P.orth)) return false;
  const int nv = M.nv, p = A.prows;
  const int nelim = P.nelim, n_red = P.n_red, nl = 3 * nelim;
  constexpr int NB = 18;                         // base + stance-leg coordinates: j < 6 base DoF j, 6 + l eliminated leg DoF l
  double* const Tm = S.RB + 32;                  // [26][6]  H(:, bl) Z
  double* const Bm = S.RB + 32 + NV * 6;         // [6][6]   Z'H(bl, bl) Z, the base block of H'
  double* const Zm = S.RB + NR * LDJ;            // [NB][6]
  double* const Cm = S.RC;
  int legd[12], Fd[NR], rowstart[4];
#pragma unroll
  for (int l = 0; l < 12; ++l) legd[l] = P.legd[l];
#pragma unroll
  for (int k = 0; k < NR; ++k) Fd[k] = P.Fd[k];
#pragma unroll
  for (int f = 0; f < 4; ++f) rowstart[f] = P.rowstart[f];
  const unsigned elimrows = P.elimrows, legrows = P.legrows;
#pragma unroll
  for (int l = 0; l < 12; ++l) asm volatile("" : "+s"(legd[l]));
#pragma unroll
  for (int k = 0; k < NR; ++k) asm volatile("" : "+s"(Fd[k]));
#pragma unroll
  for (int f = 0; f < 4; ++f) asm volatile("" : "+s"(rowstart[f]));
  int fj = 0, my_pos = -1, my_l = -1;
#pragma unroll
  for (int k = 0; k < NR; ++k) { fj = (lane == k) ? Fd[k] : fj; my_pos = (lane == Fd[k] && k < n_red) ? k : my_pos; }
#pragma unroll
  for (int l = 0; l < 12; ++l) my_l = (lane == legd[l] && l < nl) ? l : my_l;
  if (lane < 32) { S.npv[lane] = (lane < nv) ? g : 0.0; S.xv[lane] = lb; S.yv[lane] = ub; }
  OCUT(0, g + lb + ub + clb + cub);

  if (!have_h && !orth_null_basis(Cm, rowstart, legd, nelim, nl, lane, A.sing_tol, Zm, A.orth_qr != 0)) return false;
  STAMP(ts, T_P1);
  OCUT(1, Zm[lane & 63]);
  // ---- T = H(:, bl) Z: lane d + 32 h carries T[d][3 h .. 3 h + 2]
  if (!have_h) {
    const int d = lane & 31, h = lane >> 5, dd = (d < NV) ? d : NV - 1;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int dj = (j < 6) ? j : legd[j - 6];  // (rows of Z beyond 6 + nl are zero: whatever H entry they meet)
      const double hv = S.RA[dd * LDJ + dj];
      const double* zr = Zm + j * 6 + 3 * h;
      t0 = fma(hv, zr[0], t0); t1 = fma(hv, zr[1], t1); t2 = fma(hv, zr[2], t2);
      if (j % 6 == 5) __builtin_amdgcn_sched_barrier(0);   // (left alone the scheduler hoists every LDS read of the unrolled loop: 250+ VGPRs)
    }
    if (d < NV) { double* o = Tm + d * 6 + 3 * h; o[0] = t0; o[1] = t1; o[2] = t2; }
  }
  WSYNC();
  // base block of H' = Z'T(bl, :): one entry per lane (36 lanes; on the six base lanes alone the 108 FMAs + their LDS reads cost
  // the variant 90 spilled VGPRs)
  if (!have_h && lane < 36) {
    const int c = lane / 6, k = lane - 6 * c;
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int j = 0; j < NB; j += 2) {
      a0 = fma(Zm[j * 6 + c], Tm[((j < 6) ? j : legd[j - 6]) * 6 + k], a0);
      a1 = fma(Zm[(j + 1) * 6 + c], Tm[((j + 1 < 6) ? j + 1 : legd[j - 5]) * 6 + k], a1);
    }
    Bm[lane] = a0 + a1;
  }
  // per-lane column of Z (lanes >= 6: zero), kept for g' and C'
  double zcol[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) zcol[j] = (lane < 6) ? Zm[j * 6 + lane] : 0.0;
  // g' = Z'g
  double g_red = (lane >= 6 && lane < n_red) ? S.npv[fj] : 0.0;
#pragma unroll
  for (int j = 0; j < NB; ++j) g_red = fma(zcol[j], S.npv[(j < 6) ? j : legd[j - 6]], g_red);
  STAMP(ts, T_P2);
  OCUT(2, g_red + zcol[3]);
  // ---- C' = C Z for the rows that stay (in their order), then the base / stance-leg velocity bounds as the rows of Z
  double nclb = 0.0, ncub = 0.0;
  int i2 = 0;
#pragma unroll 1
  for (int i = 0; i < p; ++i) {
    if ((elimrows >> i) & 1u) continue;
    double v = (lane >= 6 && lane < n_red) ? Cm[i * LDJ + fj] : 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) v = fma(zcol[j], Cm[i * LDJ + j], v);
    if ((legrows >> i) & 1u) {
#pragma unroll
      for (int l = 0; l < 12; ++l) v = fma(zcol[6 + l], Cm[i * LDJ + legd[l]], v);
    }
    const double bl = rdl(clb, i), bu = rdl(cub, i);
    WSYNC();
    if (lane < NV) Cm[i2 * LDJ + lane] = v;
    if (lane == i2) { nclb = bl; ncub = bu; }
    WSYNC();
    ++i2;
  }
  if (cfg.use_bounds) {
    const int nb = 6 + nl;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (j < nb) { if (lane < NV) Cm[(i2 + j) * LDJ + lane] = zcol[j]; }
    }
    const int jj = lane - i2;
    if (jj >= 0 && jj < nb) {
      int dj = jj;
#pragma unroll
      for (int l = 0; l < 12; ++l) dj = (jj == 6 + l) ? legd[l] : dj;
      nclb = S.xv[dj]; ncub = S.yv[dj];
    }
    i2 += nb;
  }
  const double lb_red = (lane < 6) ? -1e30 : ((lane < n_red) ? S.xv[fj] : 0.0);
  const double ub_red = (lane < 6) ? 1e30 : ((lane < n_red) ? S.yv[fj] : 0.0);
  WSYNC();
  STAMP(ts, T_P3);
  OCUT(3, g_red + lb_red + ub_red + nclb + ncub);
  // ---- row `lane` of H' (lanes < n_red), identity padding up to NR
  if (!have_h) {
  double hr[NR];
  {
    // (every lane reads through ONE address per entry, chosen by selects: with the loads inside per-lane branches the 16 entries
    //  became 40 serialized LDS round trips)
    const int fjc = (lane < n_red) ? fj : 0;
    const double* const trow = (lane < 6) ? (Bm + lane * 6) : (Tm + fjc * 6);
    const double2a ta = lds2(trow), tb = lds2(trow + 2), tc = lds2(trow + 4);
    const double tk[6] = {ta.x, ta.y, tb.x, tb.y, tc.x, tc.y};
    const double* const src = (lane < 6) ? (Tm + lane) : (S.RA + fjc * LDJ);
    const int mul = (lane < 6) ? 6 : 1;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const double v = (k < 6) ? tk[k] : src[Fd[k] * mul];
      hr[k] = (lane < n_red && k < n_red) ? v : ((k == lane) ? 1.0 : 0.0);
    }
  }
  OCUT(4, hr[0] + hr[3] + hr[7] + hr[15] + g_red + lb_red + ub_red + nclb + ncub);
  WSYNC();
  // H' into RA rows / columns < NR — all qp_core<NR> reads of H; what is left of the 26-wide H beside it is finite and never read
  if (lane < NR) {
#pragma unroll
    for (int k = 0; k < NR; k += 2) sts2(S.RA + lane * LDJ + k, hr[k], hr[k + 1]);
  }
  WSYNC();
  }
  STAMP(ts, T_PRE);
  OCUT(5, g_red + lb_red + ub_red + nclb + ncub);
  // the reduced problem at its own compiled size (the sweeps of qp_core cost ~NM^2: n' = 11 on a 16-wide core wastes half of them)
  if (n_red <= 12) res = qp_core<12>(S, g_red, lb_red, ub_red, nclb, ncub, n_red, i2, lane, ts);
  else if (n_red <= 14) res = qp_core<14>(S, g_red, lb_red, ub_red, nclb, ncub, n_red, i2, lane, ts);
  else res = qp_core<NR>(S, g_red, lb_red, ub_red, nclb, ncub, n_red, i2, lane, ts);
  res.iters += nl + P.nlock;
  // ---- qd = Z y
  WSYNC();
  if (lane < 32) S.xv[lane] = (lane < n_red) ? res.x : 0.0;
  WSYNC();
  double x = 0.0;
  if (lane < 6 || my_l >= 0) {
    const double* zr = Zm + ((lane < 6) ? lane : 6 + my_l) * 6;
#pragma unroll
    for (int c = 0; c < 6; ++c) x = fma(zr[c], S.xv[c], x);
  } else if (my_pos >= 0) x = S.xv[my_pos];
  res.x = (lane < nv) ? x : 0.0;
  return true;
}

// ------------------------------------------------------------------------------------------------
// The orthonormal presolve without the 26-wide H: right after the task pass (At = the Cartesian task stack by DoF in LDS) the contact
// rows go through orth_qr_z, A Z is formed for the base block (one Cartesian row per lane), and H' = (A Z)'(A Z) + posture comes out of
// ONE 16 x 16 tile of the fp64 matrix cores (the full J'J is three tiles and was then reduced by Z'(H Z): 0.15 ms of the C2 step).
// Leaves H' in RA rows / columns < NR and Z at RB[16 LDJ ..) for contact_presolve_orth(.., have_h = true), which runs after the
// constraint stage. Only where the constraints are evaluated at the same state as the tasks (no second FK pass). Returns false —
// RA zeroed again for the general path — when two contact rows are numerically dependent.
// ------------------------------------------------------------------------------------------------
template <class KA>
__device__ __forceinline__ bool orth_direct_assemble(Smem& S, const KA& A, const DevModel& M, const WbcConfig& cfg,
                                                     const DevPlan& P, const double* const At, const int mtp,
                                                     const double (&lin)[3], const int lane) {
  const int nv = M.nv, nelim = P.nelim, nl = 3 * nelim, n_red = P.n_red, mc = A.mcart;
  constexpr int NB = 18;
  double* const Esc = S.RA;                      // [12][LDJ] contact rows (dead once the QR has loaded them)
  double* const AZt = S.RA;                      // [6][mtp]  (A Z)' by reduced base variable (6 mtp <= 300)
  double* const Zd = S.RA + 12 * LDJ;            // [26][6]   Z by DoF, for the A Z loop
  double* const Zs = S.RA + 12 * LDJ + NV * 6;   // [18][6]   Z by [base; stance-leg] row until At is dead
  double* const Zm = S.RB + NR * LDJ;            // ... then where contact_presolve_orth expects it
  int legd[12], Fd[NR];
#pragma unroll
  for (int l = 0; l < 12; ++l) legd[l] = P.legd[l];
#pragma unroll
  for (int k = 0; k < NR; ++k) Fd[k] = P.Fd[k];
#pragma unroll
  for (int l = 0; l < 12; ++l) asm volatile("" : "+s"(legd[l]));
#pragma unroll
  for (int k = 0; k < NR; ++k) asm volatile("" : "+s"(Fd[k]));
  // ---- the stance feet's contact rows (EEConstraint, Robot_Wrapper4.py:757-761), as the constraint stage writes them later
  {
    int fi = 0;
#pragma unroll 1
    for (unsigned cm_ = P.con_ee_mask & 15u; cm_; cm_ &= cm_ - 1) {
      const int e = __ffs((int)cm_) - 1;
      const bool sup = (lane < nv) && ((M.frame_support[WBC_FR_EE0 + e] >> lane) & 1u);
#pragma unroll
      for (int r = 0; r < 3; ++r) if (lane < NV) Esc[(3 * fi + r) * LDJ + lane] = sup ? lin[r] : 0.0;
      ++fi;
    }
  }
  WSYNC();
  const int rs4[4] = {0, 3, 6, 9};
  if (!orth_null_basis(Esc, rs4, legd, nelim, nl, lane, A.sing_tol, Zs, A.orth_qr != 0)) {
    WSYNC();
    if (lane < NV) {
#pragma unroll 1
      for (int k = 0; k < NV; k += 2) sts2(S.RA + lane * LDJ + k, 0.0, 0.0);
    }
    WSYNC();
    return false;
  }
  // ---- Z by DoF (zero rows for the DoF outside base and stance legs): the A Z loop then walks At and Z with plain strides — unrolled over
  // the [base; legs] index list its 6 accumulators x 18 terms cost the variant 150 spilled VGPRs
  {
    int zj = (lane < 6) ? lane : -1;
#pragma unroll
    for (int l = 0; l < 12; ++l) zj = (lane == legd[l] && l < nl) ? 6 + l : zj;
    if (lane < NV) {
      const double* zr = Zs + ((zj >= 0) ? zj : 0) * 6;
      const double2a z0 = lds2(zr), z1 = lds2(zr + 2), z2 = lds2(zr + 4);
      const bool on = zj >= 0;
      sts2(Zd + lane * 6, on ? z0.x : 0.0, on ? z0.y : 0.0); sts2(Zd + lane * 6 + 2, on ? z1.x : 0.0, on ? z1.y : 0.0);
      sts2(Zd + lane * 6 + 4, on ? z2.x : 0.0, on ? z2.y : 0.0);
    }
  }
  WSYNC();
  // ---- (A Z)[r][c] for the six base variables: lane = Cartesian task row r
  {
    const double* ap = At + ((lane < mc) ? lane : 0);
    const double* zp = Zd;
    double az[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll 2
    for (int d = 0; d < NV; ++d) {
      const double a = *ap;
      const double2a z0 = lds2(zp), z1 = lds2(zp + 2), z2 = lds2(zp + 4);
      az[0] = fma(a, z0.x, az[0]); az[1] = fma(a, z0.y, az[1]); az[2] = fma(a, z1.x, az[2]);
      az[3] = fma(a, z1.y, az[3]); az[4] = fma(a, z2.x, az[4]); az[5] = fma(a, z2.y, az[5]);
      ap += mtp; zp += 6;
    }
    if (lane < mc) {
#pragma unroll
      for (int c = 0; c < 6; ++c) AZt[c * mtp + lane] = az[c];
    }
  }
  WSYNC();
  // ---- H' = A_red'A_red on the matrix cores, one tile: column c0 of A_red is (A Z)[:, c0] for c0 < 6, else the column of DoF Fd[c0]
  const int kq = lane >> 4, c0 = lane & 15;
  int fdc = 0;
#pragma unroll
  for (int k = 0; k < NR; ++k) fdc = (c0 == k) ? Fd[k] : fdc;
  const double* const colp = (c0 < 6) ? (AZt + c0 * mtp) : (At + fdc * mtp);
  const bool colon = c0 < n_red;
  v4f64 acc = {0, 0, 0, 0};
#pragma unroll 1
  for (int s4 = 0; s4 < mc; s4 += 4) {
    const int r = s4 + kq;
    const double a0 = (colon && r < mc) ? colp[r] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, acc, 0, 0, 0);
  }
  WSYNC();                                       // every read of A Z and of At is done: RA rows < NR become H', RB rows >= NR take Z
  const double dp = cfg.task_joint ? (1.0 / nv) * cfg.joint_w : 0.0;   // posture rows: Z'(d^2 I)Z = d^2 I on the reduced variables
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = kq + 4 * r;
    double v = acc[r];
    if (row == c0) v = (row < n_red) ? fma(dp, dp, v) : 1.0;
    S.RA[row * LDJ + c0] = v;
  }
  if (lane < (NB * 6) / 2) sts2(Zm + 2 * lane, Zs[2 * lane], Zs[2 * lane + 1]);
  WSYNC();
  return true;
}

// ------------------------------------------------------------------------------------------------
// One instance: FK -> Jacobians -> task stack -> H, g, C, bounds [-> QP -> qdot -> q_next]
// (inputs already staged in S.in)
// ------------------------------------------------------------------------------------------------
template <int MODE, bool WARM = false, bool ORTH = false, class KA = KernelArgs>
__device__ __forceinline__ void process_instance(Smem& S, const KA& A, const DevModel& M, const WbcConfig& cfg,
                                                 const DevPlan& P, const LaneConst& lc, const InRegs& inr, const int b,
                                                 const int lane, const unsigned long long t_entry = 0) {
  const int nv = M.nv, nq = M.nq, nj = M.njoints;
  const double dt = A.dt, inv_dt = 1.0 / A.dt;   // x * (1/dt) for x / dt: one rounding more than the reference's division
  (void)dt;
  double* const oMi = S.RA + OFF_OMI;   // [joint][12]: R column-major (3 columns), then p
  const double* const qv = S.in + IN_Q;
  unsigned long long ts[T_NN];
  (void)ts;
#ifdef WBC_PROFILE
  // drain the start-up loads before the first stamp: in this build the per-phase atomics congest the memory system and
  // would otherwise be charged to the FK phase (measured: 22k of its 28k cycles). Load latency is measured on the
  // shipped build instead (bench.py with option dbg_alias_inputs).
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
  STAMP(ts, T_START);

  // ---- P1..P3: forward kinematics, frames, Jacobian columns, CoM (updateState, Robot_Wrapper4.py:400-405, 670)
  const bool need_com = cfg.task_com || cfg.con_com || (MODE == MODE_FK && (A.fk.com || A.fk.Jcom));
  FkOut fo;
  const Hdr H = load_hdr(M);
  fk_pass(S, oMi, qv, H, lc, need_com, lane, fo, ts);
  double (&lin)[3] = fo.lin; double (&ang)[3] = fo.ang; double (&com)[3] = fo.com; double (&jc)[3] = fo.jc;
  double (&Rtr)[9] = fo.Rtr; double (&ptr)[3] = fo.ptr;

  if (MODE == MODE_FK) {
    const int M0nj = A.fk_nj, M0nf = A.fk_nf;   // output strides = the largest model of the handle (mixed batches)
    // outputs of updateState: oMi / oMf (row-major R then p), data.J, com, Jcom; rows beyond this model's own count are zeroed
    if (A.fk.oMi && lane >= nj && lane < M0nj) { double* o = A.fk.oMi + ((size_t)b * M0nj + lane) * 12; for (int i = 0; i < 12; ++i) o[i] = 0.0; }
    if (A.fk.oMf && lane >= M.nframes && lane < M0nf) { double* o = A.fk.oMf + ((size_t)b * M0nf + lane) * 12; for (int i = 0; i < 12; ++i) o[i] = 0.0; }
    if (A.fk.oMi && lane < nj) {
      double* o = A.fk.oMi + ((size_t)b * M0nj + lane) * 12;
      if (lane == 0) { for (int i = 0; i < 12; ++i) o[i] = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0; }
      else {
        const double* Pj = oMi + 12 * lane;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o[3 * r + c] = Pj[3 * c + r];
        o[9] = Pj[9]; o[10] = Pj[10]; o[11] = Pj[11];
      }
    }
    if (A.fk.oMf && lane < M.nframes) {
      double* o = A.fk.oMf + ((size_t)b * M0nf + lane) * 12;
      const double* Pj = oMi + lc.fj_off;
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o[3 * r + c] = Pj[3 * c + r];
      o[9] = S.pf[3 * lane]; o[10] = S.pf[3 * lane + 1]; o[11] = S.pf[3 * lane + 2];
    }
    if (A.fk.J && lane < NV) {
      double* o = A.fk.J + (size_t)b * 6 * NV + lane;
      for (int r = 0; r < 3; ++r) { o[r * NV] = lin[r]; o[(3 + r) * NV] = ang[r]; }
    }
    if (A.fk.com && lane < 3) A.fk.com[(size_t)b * 3 + lane] = (lane == 0) ? com[0] : (lane == 1) ? com[1] : com[2];
    if (A.fk.Jcom && lane < NV) { double* o = A.fk.Jcom + (size_t)b * 3 * NV + lane; for (int r = 0; r < 3; ++r) o[r * NV] = jc[r]; }
    WSYNC();
    return;
  }
  STAMP(ts, T_FK);

  // ---- P4/P5: task stack. qpA/qpb (Robot_Wrapper4.py:1271-1294) feeding H = A'A, g = -A'b (QP_Wrapper.py:17-18)
  WSYNC();   // every lane is done reading oMi / mc: RA becomes H from here on
  double g = 0.0;
  double* const At = S.RB;                         // At[dof][row], spills over into RC (both free until P6)
  const int mtp = (A.mcart + 3) / 4 * 4 + 2;       // ≡ 2 mod 4
  int row = 0;
  if (lane < NV) {
#pragma unroll 1
    for (int k = 0; k < NV; k += 2) sts2(S.RA + lane * LDJ + k, 0.0, 0.0);
  }
  // pass 1: every lane writes its column of every Cartesian block to At and accumulates g
  // (the switches come as bit masks from the plan and each block's weights are fetched in one batch: read where they are
  //  used, every cfg.* value is its own s_load + full wait inside the dependent chain)
#pragma unroll 1
  for (unsigned tm = P.task_ee_mask; tm; tm &= tm - 1) {
    const int e = __ffs((int)tm) - 1;
    const unsigned fsup = M.frame_support[WBC_FR_EE0 + e];
    double w = cfg.ee_w[e], W0 = cfg.ee_W[e][0], W1 = cfg.ee_W[e][1], W2 = cfg.ee_W[e][2], W3 = cfg.ee_W[e][3],
           W4 = cfg.ee_W[e][4], W5 = cfg.ee_W[e][5], G0 = cfg.ee_gain[e][0], G1 = cfg.ee_gain[e][1], G2 = cfg.ee_gain[e][2];
    asm volatile("" : "+s"(w), "+s"(W0), "+s"(W1), "+s"(W2), "+s"(W3), "+s"(W4), "+s"(W5), "+s"(G0), "+s"(G1), "+s"(G2));
    const double Wd[6] = {W0, W1, W2, W3, W4, W5}, Gd[3] = {G0, G1, G2};
    const bool sup = (lane < nv) && ((fsup >> lane) & 1u);
    const double pfe[3] = {S.pf[3 * e], S.pf[3 * e + 1], S.pf[3 * e + 2]};
    double a[6];
    {  // endEffectorA2 (Robot_Wrapper4.py:474-484): LOCAL_WORLD_ALIGNED: lin + ang x p_f
      double wxp[3];
      cross3(ang, pfe, wxp);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        a[r] = sup ? Wd[r] * ((lin[r] + wxp[r]) * w) : 0.0;
        a[3 + r] = sup ? Wd[3 + r] * (ang[r] * w) : 0.0;
      }
    }
    // calcTargetVelEE3 (Robot_Wrapper4.py:1052-1157) — uniform arithmetic on the staged inputs
    const double* xt = S.in + IN_EET + 3 * e;
    const double* xp = S.in + IN_EEP + 3 * e;
    double vel[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 3; ++i) vel[i] = (xt[i] - xp[i]) * inv_dt + Gd[i] * ((xt[i] - pfe[i]) * inv_dt);
    if (A.in.ee_ref_rot) {   // omega = vee(((R* - R*_prev)/dt) R*^T)  (:1125-1128, 1133)
      const double* Rs = S.in + IN_ERR + 9 * e;
      const double* Rp = S.in + IN_EPR + 9 * e;
      double D[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) D[i] = (Rs[i] - Rp[i]) * inv_dt;
      vel[3] = D[6] * Rs[3] + D[7] * Rs[4] + D[8] * Rs[5];   // S[2][1]
      vel[4] = D[0] * Rs[6] + D[1] * Rs[7] + D[2] * Rs[8];   // S[0][2]
      vel[5] = D[3] * Rs[0] + D[4] * Rs[1] + D[5] * Rs[2];   // S[1][0]
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double br = vel[r] * w;                            // EndEffectorB2 (:907-910)
      g = fma(-a[r], br, g);
      if (lane == 0) S.bt[row + r] = br;
      if (lane < NV) At[lane * mtp + row + r] = a[r];
    }
    row += 6;
  }
  if (cfg.task_trunk) {   // trunkA (Robot_Wrapper4.py:487-490, WORLD), calcTargetVelTrunk2 (:948-1015)
    const bool sup = (lane < nv) && ((M.frame_support[WBC_FR_TRUNK] >> lane) & 1u);
    double a[6];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      a[r] = sup ? (cfg.trunk_W[r] * lin[r]) * cfg.trunk_w : 0.0;
      a[3 + r] = sup ? (cfg.trunk_W[3 + r] * ang[r]) * cfg.trunk_w : 0.0;
    }
    const double* xt = S.in + IN_TT;
    const double* xp = S.in + IN_TP;
    double vel[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) vel[i] = (xt[i] - xp[i]) * inv_dt + cfg.trunk_gain[i] * ((xt[i] - ptr[i]) * inv_dt);
    double fq[4], rq[4], Rs[9];
    R_to_quat(Rtr, fq);
    const double* er = S.in + IN_TRE;
    {
      // sin/cos of the three reference angles and of their halves: one loop body, results parked in LDS
#pragma unroll 1
      for (int i = 0; i < 6; ++i) {
        const SinCos t = sincos_cw(i < 3 ? er[i] : 0.5 * er[i - 3]);
        if (lane == 0) { S.yv[2 * i] = t.s; S.yv[2 * i + 1] = t.c; }
      }
      WSYNC();
      const double sa = S.yv[0], ca = S.yv[1], sb = S.yv[2], cb = S.yv[3], sc = S.yv[4], cc = S.yv[5];
      Rs[0] = cc * cb; Rs[1] = cc * sb * sa - sc * ca; Rs[2] = cc * sb * ca + sc * sa;
      Rs[3] = sc * cb; Rs[4] = sc * sb * sa + cc * ca; Rs[5] = sc * sb * ca - cc * sa;
      Rs[6] = -sb;     Rs[7] = cb * sa;                Rs[8] = cb * ca;
      const double qx[4] = {S.yv[6], 0, 0, S.yv[7]}, qy[4] = {0, S.yv[8], 0, S.yv[9]}, qz[4] = {0, 0, S.yv[10], S.yv[11]};
      double tq[4];
      quat_mul(qy, qx, tq);
      quat_mul(qz, tq, rq);
    }
    const double qe0 = fq[3] * rq[0] - fq[0] * rq[3] + fq[1] * rq[2] - fq[2] * rq[1];   // :974
    const double qe1 = fq[3] * rq[1] - fq[1] * rq[3] - fq[0] * rq[2] + fq[2] * rq[0];   // :975
    const double qe2 = fq[3] * rq[2] - fq[3] * rq[2] + fq[0] * rq[1] - fq[1] * rq[0];   // :976 (sic)
    const double* Ro = S.in + IN_TPR;
    double D[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) D[i] = (Rs[i] - Ro[i]) * inv_dt;
    // skew = D Rs (R*, not R*^T: :984); omega = (S[2][1], S[0][2], S[1][0]) + K qe
    vel[3] = (D[6] * Rs[1] + D[7] * Rs[4] + D[8] * Rs[7]) + cfg.trunk_gain[3] * qe0;
    vel[4] = (D[0] * Rs[2] + D[1] * Rs[5] + D[2] * Rs[8]) + cfg.trunk_gain[4] * qe1;
    vel[5] = (D[3] * Rs[0] + D[4] * Rs[3] + D[5] * Rs[6]) + cfg.trunk_gain[5] * qe2;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double br = vel[r] * cfg.trunk_w;                  // TrunkB (:914-920)
      g = fma(-a[r], br, g);
      if (lane == 0) S.bt[row + r] = br;
      if (lane < NV) At[lane * mtp + row + r] = a[r];
    }
    row += 6;
  }
  if (cfg.task_com) {     // Robot_Wrapper2 comJacobian (:600-603), cartesianTargetCoM (:661-668)
    const double* ct = S.in + IN_CT;
    const double* cv = S.in + IN_CV;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double ar = cfg.com_W[r] * jc[r];
      const double br = cv[r] + cfg.com_gain[r] * (ct[r] - com[r]);
      g = fma(-ar, br, g);
      if (lane == 0) S.bt[row + r] = br;
      if (lane < NV) At[lane * mtp + row + r] = ar;
    }
    row += 3;
  }
  WSYNC();
  STAMP(ts, T_A1);
  // pass 2: H[lane][i] = sum_r At[i][r] At[lane][r] — or, where the orthonormal contact presolve applies and the constraints are
  // evaluated at this same state, the reduced H' directly (orth_direct_assemble)
  bool direct = false;
  if (ORTH && MODE == MODE_TICK && A.presolve && A.presolve_orth && P.orth && !A.in.q_con && !(A.post_static && P.post_pert))
    direct = orth_direct_assemble(S, A, M, cfg, P, At, mtp, lin, lane);
  if (direct) {
  } else if (A.jtj_mfma) {
    // dense contraction on the fp64 matrix cores (the operand comes straight from the At image in LDS)
    const int mc = A.mcart;
    jtj_mfma(S, lane, mc, [&](int r, int c) -> double { return (r < mc && c < NV) ? At[c * mtp + r] : 0.0; });
  } else {
    // vector units, block by block over each block's DoF support (skips the structural zeros of the Jacobians)
    int r0 = 0;
#pragma unroll 1
    for (unsigned tm = P.task_ee_mask; tm; tm &= tm - 1) {
      jtj_block<6>(S, At, mtp, r0, M.frame_support[WBC_FR_EE0 + __ffs((int)tm) - 1], lane);
      r0 += 6;
    }
    if (cfg.task_trunk) { jtj_block<6>(S, At, mtp, r0, M.frame_support[WBC_FR_TRUNK], lane); r0 += 6; }
    if (cfg.task_com) { jtj_block<3>(S, At, mtp, r0, (1u << nv) - 1u, lane); r0 += 3; }
  }
  // posture rows: qpJointA (Robot_Wrapper4.py:1199-1206), qpJointb (:1209-1268)
  double dpost = 0.0, upost = 0.0;
  if (cfg.task_joint) {
    dpost = (1.0 / nv) * cfg.joint_w;
    if (cfg.task_joint == WBC_JOINT_PREV && lane < nv) upost = qv[lane < 6 ? lane : lane + 1];   // np.delete(q, 6)
    if (cfg.task_joint >= WBC_JOINT_MANI && lane < nv) {               // MANI / HYBRID (:1220-1260)
      if (A.post_static) upost = ((P.post_zero >> lane) & 1u) ? 0.0 : qv[lane < 6 ? lane : lane + 1];   // see DevPlan.post_static
      else upost = inr.pu;                                              // wbc_posture_kernel's u (or the caller's)
    }
    const double bj = (1.0 / nv) * upost * cfg.joint_w;
    if (lane < nv) g = fma(-dpost, bj, g);
    upost = bj;
  }
  if (lane < NV && !direct) S.RA[lane * LDJ + lane] += (lane < nv) ? dpost * dpost : 1.0;   // padded DoF: H_dd = 1
  if (lane >= nv) g = 0.0;
  WSYNC();

  if (MODE == MODE_ASSEMBLE) {
    const int m = A.mrows;
    if (A.qp.A && lane < NV) {
      double* o = A.qp.A + (size_t)b * m * NV;
      for (int r = 0; r < A.mcart; ++r) o[r * NV + lane] = At[lane * mtp + r];
      if (cfg.task_joint) for (int r = 0; r < NV; ++r) o[(A.mcart + r) * NV + lane] = (r == lane && lane < nv) ? dpost : 0.0;
    }
    if (A.qp.b) {
      double* o = A.qp.b + (size_t)b * m;
      if (lane < 32) for (int r = lane; r < A.mcart; r += 32) o[r] = S.bt[r];
      if (cfg.task_joint && lane < NV) o[A.mcart + lane] = (lane < nv) ? upost : 0.0;
    }
    if (A.qp.H && lane < NV) {
      double* o = A.qp.H + (size_t)b * NV * NV + (size_t)lane * NV;
      for (int k = 0; k < NV; ++k) o[k] = S.RA[lane * LDJ + k];
    }
    if (A.qp.g && lane < NV) A.qp.g[(size_t)b * NV + lane] = g;
  }
  WSYNC();   // At is dead: Cm may be written
  STAMP(ts, T_A2);
  if (A.in.q_con) {
    // qpJointb MANI/HYBRID left robot_data and current_joint_config at a perturbed configuration:
    // findConstraints, velDamperJointConstraints and integrate see THAT state. oMi scratch = RB (At is dead).
    if (lane < NQ) S.in[IN_Q + lane] = inr.qc;
    WSYNC();
    fk_pass(S, S.RB, qv, H, lc, cfg.con_com != 0, lane, fo);
    WSYNC();
  } else if (A.post_static && P.post_pert) {
    // the same state leak when every finite difference of qpJointb is structurally zero (DevPlan.post_static): each
    // perturbed entry is left at (q + d) - 2 d, and the kinematics are redone only if an active constraint depends on one
    if (lane < NQ && ((P.post_pert >> lane) & 1u)) S.in[IN_Q + lane] = (qv[lane] + 0.0002) - (0.0002 * 2);
    WSYNC();
    if (P.post_fk2) { fk_pass(S, S.RB, qv, H, lc, cfg.con_com != 0, lane, fo); WSYNC(); }
  }

  // ---- P6: constraints in order CoM, Trunk, FR, FL, RR, RL, Grip: findConstraints (Robot_Wrapper4.py:764-836)
  double* const Cm = S.RC;
  double clb = 0.0, cub = 0.0;
  int prow = 0;
  if (cfg.con_com) {   // CoMConstraint (Robot_Wrapper4.py:669-694); EE_frame_pos[1] = FL, [2] = RR
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (lane < NV) Cm[(prow + r) * LDJ + lane] = jc[r];
      const double lo = ((S.pf[3 * 2 + r] - com[r]) * inv_dt) * cfg.com_box_scale;
      const double hi = ((S.pf[3 * 1 + r] - com[r]) * inv_dt) * cfg.com_box_scale;
      if (lane == prow + r) { clb = lo; cub = hi; }
    }
    prow += 2;
  }
  if (cfg.con_trunk) { // trunkConstraint (Robot_Wrapper4.py:707-754): LOCAL_WORLD_ALIGNED rows z, wx, wy, wz
    const bool sup = (lane < nv) && ((M.frame_support[WBC_FR_TRUNK] >> lane) & 1u);
    double wxp[3];
    cross3(ang, ptr, wxp);
    const double rowv[4] = {sup ? lin[2] + wxp[2] : 0.0, sup ? ang[0] : 0.0, sup ? ang[1] : 0.0, sup ? ang[2] : 0.0};
    const double* bc = S.in + IN_BOX;
    // scipy as_euler('xyz') of the trunk rotation (:714-715): roll = atan2(R21, R22), pitch = -asin(R20) =
    // atan2(-R20, |(R21, R22)|), yaw = atan2(R10, R00) — ONE atan2 evaluated on lanes 0..2, then broadcast
    const double ay = (lane == 0) ? Rtr[7] : ((lane == 1) ? -Rtr[6] : Rtr[3]);
    const double ax = (lane == 0) ? Rtr[8] : ((lane == 1) ? sqrt(fma(Rtr[7], Rtr[7], Rtr[8] * Rtr[8])) : Rtr[0]);
    const double eul = atan2(ay, ax);
    const double cur[4] = {ptr[2], rdl(eul, 0), rdl(eul, 1), rdl(eul, 2)};
#pragma unroll
    for (int r = 0; r < 4; ++r) if (lane < NV) Cm[(prow + r) * LDJ + lane] = rowv[r];
    {   // the lane that owns row prow + r computes that row's bounds (:719-736)
      const int r = lane - prow;
      const double bcr = (r == 0) ? bc[0] : (r == 1) ? bc[1] : (r == 2) ? bc[2] : bc[3];
      const double cr = (r == 0) ? cur[0] : (r == 1) ? cur[1] : (r == 2) ? cur[2] : cur[3];
      const double vr = (r == 0) ? bc[0] * cfg.trunk_box_z_frac : cfg.trunk_box_ang;
      if (r >= 0 && r < 4) {
        clb = (((bcr - vr) - cr) * inv_dt) * cfg.trunk_box_scale;   // :735
        cub = (((bcr + vr) - cr) * inv_dt) * cfg.trunk_box_scale;   // :736
      }
    }
    prow += 4;
  }
#pragma unroll 1
  for (unsigned cm_ = P.con_ee_mask; cm_; cm_ &= cm_ - 1) {   // EEConstraint (Robot_Wrapper4.py:757-761): WORLD rows 0..2, 0 <= . <= 0
    const int e = __ffs((int)cm_) - 1;
    const bool sup = (lane < nv) && ((M.frame_support[WBC_FR_EE0 + e] >> lane) & 1u);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      if (lane < NV) Cm[(prow + r) * LDJ + lane] = sup ? lin[r] : 0.0;
      if (lane == prow + r) { clb = 0.0; cub = 0.0; }
    }
    prow += 3;
  }
  STAMP(ts, T_A3);
  // ---- velDamperJointConstraints (Robot_Wrapper4.py:572-637), index map from cfg
  double lb = 0.0, ub = 0.0;
  if (lane < nv) {
    if (!cfg.use_bounds) { lb = -1e30; ub = 1e30; }
    else {
      const double qi = qv[lc.dq_idx], lo = lc.d_lo, hi = lc.d_hi, vm = lc.d_vm;
      if (qi <= lo + cfg.damper_qi) {
        lb = -cfg.damper_coef * (qi - lo - cfg.damper_qs) / (cfg.damper_qi - cfg.damper_qs);
        if (lb > vm) lb = vm;
        if (lb < -vm) lb = -vm;
      } else lb = -vm;
      if (qi >= hi - cfg.damper_qi) {
        ub = cfg.damper_coef * (hi - qi - cfg.damper_qs) / (cfg.damper_qi - cfg.damper_qs);
        if (ub < -vm) ub = -vm;
        if (ub > vm) ub = vm;
      } else ub = vm;
      if (lb > 0) lb = -lb;
      if (ub < 0) ub = -ub;
      if (lane >= cfg.lock_from) { lb = 0.0; ub = 0.0; }
    }
  }
  WSYNC();
  if (MODE == MODE_ASSEMBLE) {
    const int p = A.prows;
    if (A.qp.C && lane < NV) { double* o = A.qp.C + (size_t)b * p * NV; for (int r = 0; r < p; ++r) o[r * NV + lane] = Cm[r * LDJ + lane]; }
    if (A.qp.Clb && lane < p) A.qp.Clb[(size_t)b * p + lane] = clb;
    if (A.qp.Cub && lane < p) A.qp.Cub[(size_t)b * p + lane] = cub;
    if (A.qp.lb && lane < NV) A.qp.lb[(size_t)b * NV + lane] = lb;
    if (A.qp.ub && lane < NV) A.qp.ub[(size_t)b * NV + lane] = ub;
    WSYNC();
    return;
  }
  STAMP(ts, T_ASM);

  // ---- P7/P8: the QP (QP_Wrapper.py:23-73). Padded DoF (lane >= nv) carry no constraint and stay 0.
  QpResult res;
#ifdef WBC_PROFILE
  ts[T_PRE] = 0;
#endif
  // warm start in the problem's own indexing: lane d <-> bound of DoF d, lane i <-> constraint row i
  const unsigned long long w0 = (WARM && A.ws_in) ? A.ws_in[2 * (size_t)b] : 0ull, w1 = (WARM && A.ws_in) ? A.ws_in[2 * (size_t)b + 1] : 0ull;
  if (!contact_presolve<WARM>(S, A, M, cfg, P, dpost, g, lb, ub, clb, cub, lane, ts, res, w0, w1) &&
      !(ORTH && contact_presolve_orth(S, A, M, cfg, P, g, lb, ub, clb, cub, lane, ts, res, direct))) {
    const int sb = (lane < 32) ? (int)(((w0 >> lane) & 1ull) | (((w0 >> (32 + lane)) & 1ull) << 1)) : 0;
    const int sr = (lane < 32) ? (int)(((w1 >> lane) & 1ull) | (((w1 >> (32 + lane)) & 1ull) << 1)) : 0;
    // The DoF the velocity box locks at 0 (>= lock_from, Robot_Wrapper4.py:627-630) are the LAST ones: they leave the problem (x = 0
    // contributes to nothing) and the solve runs on a 24-wide core where that fits — the sweeps of qp_core cost ~NM^2. Counted as
    // working-set changes so that `iters` keeps its meaning.
    const int ntail = (!WARM && cfg.use_bounds && cfg.lock_from >= 6 && cfg.lock_from < nv) ? nv - cfg.lock_from : 0;
    const int n_eff = nv - ntail;
    if (ntail > 0 && n_eff <= 24) {
      WSYNC();
#pragma unroll 1
      for (int k = n_eff; k < 24; ++k) {           // identity padding of H, zero columns of C
        if (lane < 24) { S.RA[lane * LDJ + k] = (lane == k) ? 1.0 : 0.0; if (lane != k) S.RA[k * LDJ + lane] = 0.0; }
        if (lane < A.prows) S.RC[lane * LDJ + k] = 0.0;
      }
      WSYNC();
      res = qp_core<24>(S, (lane < n_eff) ? g : 0.0, lb, ub, clb, cub, n_eff, A.prows, lane, ts);
      res.iters += ntail;
      if (lane >= n_eff) res.x = 0.0;
    } else
    res = qp_core<NV, Smem, LDJ, WARM>(S, g, lb, ub, clb, cub, nv, A.prows, lane, ts, 0, sb == 3 ? 0 : sb, sr == 3 ? 0 : sr);
  }
  if (WARM && A.ws_out) {   // (res.ws_* are in full-problem indexing on every path; an unsolved QP carries nothing)
    const unsigned long long o0 = (__ballot(res.ws_b == 1) & 0xFFFFFFFFull) | (__ballot(res.ws_b == 2) << 32);
    const unsigned long long o1 = (__ballot(res.ws_r == 1) & 0xFFFFFFFFull) | (__ballot(res.ws_r == 2) << 32);
    if (lane == 0) { A.ws_out[2 * (size_t)b] = o0; A.ws_out[2 * (size_t)b + 1] = o1; }
  }
  if (A.out.qdot && lane < NV) A.out.qdot[(size_t)b * NV + lane] = (lane < nv) ? res.x : 0.0;
  if (lane == 0) {
    if (A.out.status) A.out.status[b] = res.status;
    if (A.out.iters) A.out.iters[b] = res.iters;
  }
  // ---- jointVelocitiestoConfig (Robot_Wrapper4.py:440-441): q_next = pin.integrate(q, qdot * dt)
  if (A.out.q_next) {
    double* qn = A.out.q_next + (size_t)b * NQ;
    const double v = res.x * dt;
    if (lane < 32) S.xv[lane] = (lane < nv) ? v : 0.0;
    WSYNC();
    integrate_ff(S, lane, qn);
    if (lane >= 6 && lane < nv) qn[lc.col_q] = qv[lc.col_q] + v;
    if (lane >= nq && lane < NQ) qn[lane] = 0.0;
    WSYNC();
  }
#ifdef WBC_PROFILE
  STAMP(ts, T_END);
  if (A.prof && lane == 0 && res.status == WBC_QP_OPTIMAL) {
    for (int i = 1; i < T_N; ++i) atomicAdd(A.prof + i, ts[i] - ts[i - 1]);
    atomicAdd(A.prof + 0, 1ull);
    atomicAdd(A.prof + 8, (unsigned long long)res.iters);
    atomicAdd(A.prof + 9, ts[T_A1] - ts[T_FK]);     // task rows + targets
    atomicAdd(A.prof + 10, ts[T_A2] - ts[T_A1]);    // J'J + posture
    atomicAdd(A.prof + 11, ts[T_A3] - ts[T_A2]);    // constraint rows (incl. trunk Euler angles)
    atomicAdd(A.prof + 12, ts[T_ASM] - ts[T_A3]);   // damper bounds
    if (ts[T_PRE]) {   // contact presolve (inside [3]): total, engaged count, then G / H' g' / C' rows / H' store
      atomicAdd(A.prof + 13, ts[T_PRE] - ts[T_ASM]); atomicAdd(A.prof + 14, 1ull);
      atomicAdd(A.prof + 16, ts[T_P1] - ts[T_ASM]); atomicAdd(A.prof + 17, ts[T_P2] - ts[T_P1]);
      atomicAdd(A.prof + 18, ts[T_P3] - ts[T_P2]); atomicAdd(A.prof + 19, ts[T_PRE] - ts[T_P3]);
    }
    atomicAdd(A.prof + 23, ts[T_START] - t_entry);   // kernel entry -> inputs staged (load latency)
    atomicAdd(A.prof + 20, ts[T_F1] - ts[T_START]); atomicAdd(A.prof + 21, ts[T_F2] - ts[T_F1]); atomicAdd(A.prof + 22, ts[T_FK] - ts[T_F2]);
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// kernels: single-wave workgroups, one per instance for the tick kernels (the QP / integrate kernels walk the batch with a
// grid-stride loop whose exit, b >= B, every wave reaches).
// ------------------------------------------------------------------------------------------------
// WARM: the variant that reads / writes working sets (warm start, KernelArgs.ws_in / ws_out); the cold variant carries none of it
// ORTH: the variant that carries contact_presolve_orth (chosen by launch_tick when a plan of the batch asks for it: the other
// variants keep their register allocation — with the extra code inlined the general kernel went from 198 VGPRs to 256 + spills)
template <int MODE, bool WARM = false, bool ORTH = false>
__global__ void __launch_bounds__(64, 2) wbc_tick_kernel(const KernelArgs A, const DevModel* __restrict__ models,
                                                         const WbcConfig* __restrict__ cfgs, const DevPlan* __restrict__ plans) {
  // models / cfgs are separate __restrict__ const parameters so that the compiler may read them with scalar loads
  // (as members of A it must assume the kernel's own stores clobber them: every access became a vector load + full wait).
  // ONE instance per single-wave workgroup, no loop: inside a persistent loop the compiler hoists hundreds of
  // "invariants" (polynomial coefficients, masks, addresses) out of the tick, spills them to scratch and reloads them
  // one by one with full memory waits (profiles/r01_phase_cycles_v6: 60k cycles in one atan2). The hardware's
  // workgroup dispatcher does the batch loop instead; other resident waves cover this wave's input latency.
  __shared__ Smem S;
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
#ifdef WBC_PROFILE
  const unsigned long long t_entry = clock64();
#else
  const unsigned long long t_entry = 0;
#endif
  S.cl[lane] = 0.0;                            // zero padding the rotating loops rely on (never written above entry 25)
  const bool has2 = A.in.trunk_target || A.in.prev_trunk_target || A.in.trunk_ref_euler || A.in.trunk_prev_rot ||
                    A.in.com_target || A.in.com_target_vel;
  const bool has3 = A.in.ee_ref_rot != nullptr;
  // the model index is wave-uniform: say so, or every M.* / cfg.* access becomes a vector load
  const int mid = model_index(A.in.model_id, b, A.n_models);
  const InRegs cur = load_inputs(A.in, A.dbg_alias ? 0 : b, lane, has2, has3);   // dbg_alias: diagnostic, every wave reads instance 0
  const LaneConst lc = load_lane_const(models[mid], cfgs[mid], lane);   // L1/L2-resident 3 KB table
  stage_inputs(S, cur, lane, has2, has3);
  WSYNC();
  process_instance<MODE, WARM, ORTH>(S, A, models[mid], cfgs[mid], plans[mid], lc, cur, b, lane, t_entry);
}

// ------------------------------------------------------------------------------------------------
// The sim3-tick kernel (compact LDS, 12 workgroups per CU instead of 8): for batches whose every (model, configuration)
// has a structural presolve plan (DevPlan.enabled: no task touches the stance legs), at most WBC_SIM3_MAXP constraint
// rows and no orientation references. Same arithmetic as process_instance + contact_presolve, but the QP is assembled
// directly in its reduced form: the task stack is stored by REDUCED variable (At[k][row], k = DevPlan.pos[dof]), H' is
// accumulated as a 16 x 16 matrix, the original constraint rows pass through a scratch image (RB) from which G and
// C' = C Z are read. Only qp_core<16> is compiled in. An instance whose presolve cannot be applied (singular leg block)
// is marked WBC_QP_DEFERRED and redone by the general kernel in an early-exit second pass.
// ------------------------------------------------------------------------------------------------
constexpr int CSC = 18;                      // row stride of the reduced constraint matrix (18 = 2 mod 4, >= NR)
constexpr int PC = WBC_SIM3_MAXP;            // rows of the reduced constraint matrix (kept rows + leg-bound rows)
struct __attribute__((aligned(16))) SmemC {
  double RA[NR * LDJ];                  // oMi + m c (FK) -> H' -> B columns -> T
  double RB[NR * LDJ];                  // At[k][row] (task stack by reduced variable) -> second-pass oMi -> original C rows -> J
  double RC[PC * CSC];                  // C' (reduced constraint rows)
  double in[128];                       // this instance's inputs (groups 1 and 2)
  double pf[WBC_MAX_FRAMES * 3];
  double dv[32], xv[32], npv[32], lv[32], dinv[32], yv[32];
  double cl[48];                        // Cholesky column broadcast; entries 16..47 stay zero
  double Gm[12 * GS];                   // G: eliminated leg DoF l (row) x [base DoF | extra unknowns of the pivoted feet] (column)
};

// H'[lane][k] += sum_r At[k][row0 + r] At[lane][row0 + r] for the reduced variables k in `mask`
template <int NRW>
__device__ __forceinline__ void jtj_block_c(SmemC& S, const double* At, const int mtp, const int row0, unsigned mask,
                                            const int lane) {
  const int li = lane < NR ? lane : NR - 1;
  double a[NRW];
#pragma unroll
  for (int r = 0; r < NRW; ++r) a[r] = At[li * mtp + row0 + r];
#pragma unroll 1
  while (mask) {
    const int i0 = __ffs((int)mask) - 1;
    mask &= mask - 1;
    const bool two = mask != 0;
    const int i1 = two ? __ffs((int)mask) - 1 : i0;
    mask &= mask - 1;
    double s0 = S.RA[li * LDJ + i0], s1 = S.RA[li * LDJ + i1];
#pragma unroll
    for (int r = 0; r < NRW; ++r) { s0 = fma(At[i0 * mtp + row0 + r], a[r], s0); s1 = fma(At[i1 * mtp + row0 + r], a[r], s1); }
    if (lane < NR) { S.RA[lane * LDJ + i0] = s0; if (two) S.RA[lane * LDJ + i1] = s1; }
  }
}

// Ablation timing (option "dbg_stop", diagnostic build): the sim3 kernel ends after stage k with a store that keeps the stage's
// results alive; run time of stage k = T(stop k) - T(stop k - 1). Stages: 1 FK + Jacobian columns, 2 task rows, 3 J'J +
// posture, 4 constraint rows + damper bounds, 5 presolve (G, g', C', H'), 6 Cholesky / substitutions, 7 equality phase,
// 0 = the whole tick.
// Compiled in only with -DWBC_ABLATE (csrc/Makefile target `ablate`): in the shipped kernel the stores that keep a cut stage's
// results alive cost 24 spilled VGPRs (0.648 -> 0.688 ms per 65536 ticks), so there the macro is empty.
#ifdef WBC_ABLATE
#define DBG_STOP(k, val) do { if (A.dbg_stop == (k)) { if (lane < NV) A.out.qdot[(size_t)b * NV + lane] = (val); \
                                                      if (lane == 0) A.out.status[b] = 0; return; } } while (0)
#define DBG_STOP_ARG A.dbg_stop
#else
#define DBG_STOP(k, val) do { } while (0)
#define DBG_STOP_ARG 0
#endif

template <bool WARM>
__device__ __forceinline__ void process_sim3(SmemC& S, const KernelArgs& A, const DevModel& M, const WbcConfig& cfg,
                                             const DevPlan& P, const Hdr& H, const LaneConst& lc, const InRegs& inr,
                                             const int b, const int lane, const unsigned long long ws0, const unsigned long long ws1) {
  const int nv = H.nv, nq = H.nq;
  // the configuration's switches as ONE word from the plan (each cfg.* flag read where it is tested costs its own scalar load +
  // full wait): bit 0 con_com, 1 con_trunk, 2 task_trunk, 3 use_bounds, bits 4..6 task_joint
  const unsigned fl = P.flags;
  const bool c_con_com = fl & 1u, c_con_trunk = fl & 2u, c_task_trunk = fl & 4u, c_use_bounds = fl & 8u;
  const int c_task_joint = (fl >> 4) & 7u;
  const double dt = A.dt, inv_dt = 1.0 / A.dt;
  const double* const qv = S.in + IN_Q;
  unsigned long long ts[T_NN];
  (void)ts;
#ifdef WBC_PROFILE
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // see process_instance: start-up loads are not charged to FK
#endif
  STAMP(ts, T_START);
  // ---- P1..P3 (updateState, Robot_Wrapper4.py:400-405)
  FkOut fo;
  fk_pass(S, S.RA, qv, H, lc, c_con_com, lane, fo);
  double (&lin)[3] = fo.lin; double (&ang)[3] = fo.ang; double (&com)[3] = fo.com; double (&jc)[3] = fo.jc;
  double (&Rtr)[9] = fo.Rtr; double (&ptr)[3] = fo.ptr;

  STAMP(ts, T_FK);
  DBG_STOP(1, lin[0] + ang[1] + ptr[2] + Rtr[4]);
  // ---- the plan's index maps: one batch of scalar loads, per-lane views by select chains
  const int nelim = P.nelim, n_red = P.n_red, nl = 3 * nelim, p_keep = P.p_keep, p = A.prows;
  int legd[12], Fd[NR], rowstart[4];
#pragma unroll
  for (int l = 0; l < 12; ++l) legd[l] = P.legd[l];
#pragma unroll
  for (int k = 0; k < NR; ++k) Fd[k] = P.Fd[k];
#pragma unroll
  for (int f = 0; f < 4; ++f) rowstart[f] = P.rowstart[f];
  const unsigned elimrows = P.elimrows, legrows = P.legrows;
#pragma unroll
  for (int l = 0; l < 12; ++l) asm volatile("" : "+s"(legd[l]));
#pragma unroll
  for (int k = 0; k < NR; ++k) asm volatile("" : "+s"(Fd[k]));
#pragma unroll
  for (int f = 0; f < 4; ++f) asm volatile("" : "+s"(rowstart[f]));
  int fj = 0, my_pos = -1, my_l = -1, my_legd = 0;
#pragma unroll
  for (int k = 0; k < NR; ++k) { fj = (lane == k) ? Fd[k] : fj; my_pos = (lane == Fd[k] && k < n_red) ? k : my_pos; }
#pragma unroll
  for (int l = 0; l < 12; ++l) { my_l = (lane == legd[l] && l < nl) ? l : my_l; my_legd = (lane - p_keep == l) ? legd[l] : my_legd; }

  // ---- task stack, pass 1 (qpA / qpb, Robot_Wrapper4.py:1271-1294): lane = DoF; its column goes to At[pos][row]
  WSYNC();   // every lane is done reading oMi: RA becomes H'
  double g = 0.0;
  double* const At = S.RB;
  const int mtp = (A.mcart + 3) / 4 * 4 + 2;
  int row = 0;
  if (lane < NR) {
#pragma unroll
    for (int k = 0; k < NR; k += 2) sts2(S.RA + lane * LDJ + k, 0.0, 0.0);
  }
  const bool stores = my_pos >= 0;
  const int arow = (stores ? my_pos : 0) * mtp;
  // (the switches come as bit masks from the plan and each block's weights are fetched in one batch: read where they are
  //  used, every cfg.* value is its own s_load + full wait inside the dependent chain)
#pragma unroll 1
  for (unsigned tm = P.task_ee_mask; tm; tm &= tm - 1) {
    const int e = __ffs((int)tm) - 1;
    const unsigned fsup = M.frame_support[WBC_FR_EE0 + e];
    double w = cfg.ee_w[e], W0 = cfg.ee_W[e][0], W1 = cfg.ee_W[e][1], W2 = cfg.ee_W[e][2], W3 = cfg.ee_W[e][3],
           W4 = cfg.ee_W[e][4], W5 = cfg.ee_W[e][5], G0 = cfg.ee_gain[e][0], G1 = cfg.ee_gain[e][1], G2 = cfg.ee_gain[e][2];
    asm volatile("" : "+s"(w), "+s"(W0), "+s"(W1), "+s"(W2), "+s"(W3), "+s"(W4), "+s"(W5), "+s"(G0), "+s"(G1), "+s"(G2));
    const double Wd[6] = {W0, W1, W2, W3, W4, W5}, Gd[3] = {G0, G1, G2};
    const bool sup = (lane < nv) && ((fsup >> lane) & 1u);
    const double pfe[3] = {S.pf[3 * e], S.pf[3 * e + 1], S.pf[3 * e + 2]};
    double a[6];
    {  // endEffectorA2 (Robot_Wrapper4.py:474-484): LOCAL_WORLD_ALIGNED: lin + ang x p_f
      double wxp[3];
      cross3(ang, pfe, wxp);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        a[r] = sup ? Wd[r] * ((lin[r] + wxp[r]) * w) : 0.0;
        a[3 + r] = sup ? Wd[3 + r] * (ang[r] * w) : 0.0;
      }
    }
    const double* xt = S.in + IN_EET + 3 * e;
    const double* xp = S.in + IN_EEP + 3 * e;
    double vel[6] = {0, 0, 0, 0, 0, 0};   // calcTargetVelEE3 (:1052-1157) with R* == R*_prev (no orientation references here)
#pragma unroll
    for (int i = 0; i < 3; ++i) vel[i] = (xt[i] - xp[i]) * inv_dt + Gd[i] * ((xt[i] - pfe[i]) * inv_dt);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double br = vel[r] * w;                            // EndEffectorB2 (:907-910)
      g = fma(-a[r], br, g);
      if (stores) At[arow + row + r] = a[r];
    }
    row += 6;
  }
  if (c_task_trunk) {   // trunkA (Robot_Wrapper4.py:487-490, WORLD), calcTargetVelTrunk2 (:948-1015)
    const bool sup = (lane < nv) && ((M.frame_support[WBC_FR_TRUNK] >> lane) & 1u);
    double a[6];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      a[r] = sup ? (cfg.trunk_W[r] * lin[r]) * cfg.trunk_w : 0.0;
      a[3 + r] = sup ? (cfg.trunk_W[3 + r] * ang[r]) * cfg.trunk_w : 0.0;
    }
    const double* xt = S.in + IN_TT;
    const double* xp = S.in + IN_TP;
    double vel[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) vel[i] = (xt[i] - xp[i]) * inv_dt + cfg.trunk_gain[i] * ((xt[i] - ptr[i]) * inv_dt);
    double fq[4], rq[4], Rs[9];
    R_to_quat(Rtr, fq);
    const double* er = S.in + IN_TRE;
    {
#pragma unroll 1
      for (int i = 0; i < 6; ++i) {
        const SinCos t = sincos_cw(i < 3 ? er[i] : 0.5 * er[i - 3]);
        if (lane == 0) { S.yv[2 * i] = t.s; S.yv[2 * i + 1] = t.c; }
      }
      WSYNC();
      const double sa = S.yv[0], ca = S.yv[1], sb = S.yv[2], cb = S.yv[3], sc = S.yv[4], cc = S.yv[5];
      Rs[0] = cc * cb; Rs[1] = cc * sb * sa - sc * ca; Rs[2] = cc * sb * ca + sc * sa;
      Rs[3] = sc * cb; Rs[4] = sc * sb * sa + cc * ca; Rs[5] = sc * sb * ca - cc * sa;
      Rs[6] = -sb;     Rs[7] = cb * sa;                Rs[8] = cb * ca;
      const double qx[4] = {S.yv[6], 0, 0, S.yv[7]}, qy[4] = {0, S.yv[8], 0, S.yv[9]}, qz[4] = {0, 0, S.yv[10], S.yv[11]};
      double tq[4];
      quat_mul(qy, qx, tq);
      quat_mul(qz, tq, rq);
    }
    const double qe0 = fq[3] * rq[0] - fq[0] * rq[3] + fq[1] * rq[2] - fq[2] * rq[1];   // :974
    const double qe1 = fq[3] * rq[1] - fq[1] * rq[3] - fq[0] * rq[2] + fq[2] * rq[0];   // :975
    const double qe2 = fq[3] * rq[2] - fq[3] * rq[2] + fq[0] * rq[1] - fq[1] * rq[0];   // :976 (sic)
    const double* Ro = S.in + IN_TPR;
    double D[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) D[i] = (Rs[i] - Ro[i]) * inv_dt;
    vel[3] = (D[6] * Rs[1] + D[7] * Rs[4] + D[8] * Rs[7]) + cfg.trunk_gain[3] * qe0;
    vel[4] = (D[0] * Rs[2] + D[1] * Rs[5] + D[2] * Rs[8]) + cfg.trunk_gain[4] * qe1;
    vel[5] = (D[3] * Rs[0] + D[4] * Rs[3] + D[5] * Rs[6]) + cfg.trunk_gain[5] * qe2;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double br = vel[r] * cfg.trunk_w;                  // TrunkB (:914-920)
      g = fma(-a[r], br, g);
      if (stores) At[arow + row + r] = a[r];
    }
    row += 6;
  }
  if (lane >= n_red && lane < NR) {   // padded reduced variables carry no task rows
#pragma unroll 1
    for (int r = 0; r < A.mcart; ++r) At[lane * mtp + r] = 0.0;
  }
  WSYNC();
  STAMP(ts, T_A1);
  DBG_STOP(2, g + At[(lane & 15) * mtp]);
  // ---- pass 2: H'[lane][k] = sum_r At[k][r] At[lane][r], block by block over each block's reduced support
  {
    int r0 = 0;
#pragma unroll 1
    for (unsigned tm = P.task_ee_mask; tm; tm &= tm - 1) {
      jtj_block_c<6>(S, At, mtp, r0, P.redsup[WBC_FR_EE0 + __ffs((int)tm) - 1], lane);
      r0 += 6;
    }
    if (c_task_trunk) { jtj_block_c<6>(S, At, mtp, r0, P.redsup[WBC_FR_TRUNK], lane); r0 += 6; }
  }
  // posture rows: qpJointA (Robot_Wrapper4.py:1199-1206), qpJointb (:1209-1268); lane = DoF
  const double dpost = (1.0 / nv) * cfg.joint_w;
  {
    double upost = 0.0;
    if (c_task_joint == WBC_JOINT_PREV && lane < nv) upost = qv[lane < 6 ? lane : lane + 1];
    if (c_task_joint >= WBC_JOINT_MANI && lane < nv) {
      if (A.post_static) upost = ((P.post_zero >> lane) & 1u) ? 0.0 : qv[lane < 6 ? lane : lane + 1];   // see DevPlan.post_static
      else upost = inr.pu;
    }
    const double bj = (1.0 / nv) * upost * cfg.joint_w;
    if (lane < nv) g = fma(-dpost, bj, g);
  }
  if (lane >= nv) g = 0.0;
  if (lane < NR) S.RA[lane * LDJ + lane] += (lane < n_red) ? dpost * dpost : 1.0;
  WSYNC();   // At is dead: RB may be reused
  STAMP(ts, T_A2);
  DBG_STOP(3, g + S.RA[(lane & 15) * LDJ + 3]);
  if (A.in.q_con) {   // the configuration qpJointb MANI/HYBRID left behind: constraints, bounds, integrate see it
    if (lane < NQ) S.in[IN_Q + lane] = inr.qc;
    WSYNC();
    fk_pass(S, S.RB, qv, H, lc, c_con_com, lane, fo);
    WSYNC();
  } else if (A.post_static && P.post_pert) {   // same leak, structurally-zero gradients (see process_instance)
    if (lane < NQ && ((P.post_pert >> lane) & 1u)) S.in[IN_Q + lane] = (qv[lane] + 0.0002) - (0.0002 * 2);
    WSYNC();
    if (P.post_fk2) { fk_pass(S, S.RB, qv, H, lc, c_con_com, lane, fo); WSYNC(); }
  }

  // ---- original constraint rows (findConstraints order, Robot_Wrapper4.py:764-836) into the scratch image Co = RB [p][26]
  double* const Co = S.RB;
  double clb = 0.0, cub = 0.0;
  int prow = 0;
  if (c_con_com) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (lane < NV) Co[(prow + r) * LDJ + lane] = jc[r];
      const double lo = ((S.pf[3 * 2 + r] - com[r]) * inv_dt) * cfg.com_box_scale;
      const double hi = ((S.pf[3 * 1 + r] - com[r]) * inv_dt) * cfg.com_box_scale;
      if (lane == prow + r) { clb = lo; cub = hi; }
    }
    prow += 2;
  }
  if (c_con_trunk) {
    const bool sup = (lane < nv) && ((M.frame_support[WBC_FR_TRUNK] >> lane) & 1u);
    double wxp[3];
    cross3(ang, ptr, wxp);
    const double rowv[4] = {sup ? lin[2] + wxp[2] : 0.0, sup ? ang[0] : 0.0, sup ? ang[1] : 0.0, sup ? ang[2] : 0.0};
    const double* bc = S.in + IN_BOX;
    const double ay = (lane == 0) ? Rtr[7] : ((lane == 1) ? -Rtr[6] : Rtr[3]);
    const double ax = (lane == 0) ? Rtr[8] : ((lane == 1) ? sqrt(fma(Rtr[7], Rtr[7], Rtr[8] * Rtr[8])) : Rtr[0]);
    const double eul = atan2(ay, ax);
    const double cur[4] = {ptr[2], rdl(eul, 0), rdl(eul, 1), rdl(eul, 2)};
#pragma unroll
    for (int r = 0; r < 4; ++r) if (lane < NV) Co[(prow + r) * LDJ + lane] = rowv[r];
    {
      const int r = lane - prow;
      const double bcr = (r == 0) ? bc[0] : (r == 1) ? bc[1] : (r == 2) ? bc[2] : bc[3];
      const double cr = (r == 0) ? cur[0] : (r == 1) ? cur[1] : (r == 2) ? cur[2] : cur[3];
      const double vr = (r == 0) ? bc[0] * cfg.trunk_box_z_frac : cfg.trunk_box_ang;
      if (r >= 0 && r < 4) {
        clb = (((bcr - vr) - cr) * inv_dt) * cfg.trunk_box_scale;
        cub = (((bcr + vr) - cr) * inv_dt) * cfg.trunk_box_scale;
      }
    }
    prow += 4;
  }
#pragma unroll 1
  for (unsigned cm_ = P.con_ee_mask; cm_; cm_ &= cm_ - 1) {
    const int e = __ffs((int)cm_) - 1;
    const bool sup = (lane < nv) && ((M.frame_support[WBC_FR_EE0 + e] >> lane) & 1u);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      if (lane < NV) Co[(prow + r) * LDJ + lane] = sup ? lin[r] : 0.0;
      if (lane == prow + r) { clb = 0.0; cub = 0.0; }
    }
    prow += 3;
  }
  STAMP(ts, T_A3);
  // ---- velDamperJointConstraints (Robot_Wrapper4.py:572-637), lane = DoF
  double lb = 0.0, ub = 0.0;
  if (lane < nv) {
    if (!c_use_bounds) { lb = -1e30; ub = 1e30; }
    else {
      const double qi = qv[lc.dq_idx], lo = lc.d_lo, hi = lc.d_hi, vm = lc.d_vm;
      if (qi <= lo + cfg.damper_qi) {
        lb = -cfg.damper_coef * (qi - lo - cfg.damper_qs) / (cfg.damper_qi - cfg.damper_qs);
        if (lb > vm) lb = vm;
        if (lb < -vm) lb = -vm;
      } else lb = -vm;
      if (qi >= hi - cfg.damper_qi) {
        ub = cfg.damper_coef * (hi - qi - cfg.damper_qs) / (cfg.damper_qi - cfg.damper_qs);
        if (ub < -vm) ub = -vm;
        if (ub > vm) ub = vm;
      } else ub = vm;
      if (lb > 0) lb = -lb;
      if (ub < 0) ub = -ub;
      if (lane >= cfg.lock_from) { lb = 0.0; ub = 0.0; }
    }
  }
  if (lane < 32) { S.npv[lane] = g; S.xv[lane] = lb; S.yv[lane] = ub; }
  WSYNC();
  STAMP(ts, T_ASM);
  DBG_STOP(4, g + lb + ub + clb + cub + Co[(lane & 15) * LDJ + 2]);

  // ---- G_e = -K_e^-1 B_e, all feet at once (see contact_presolve)
  double* const Gm = S.Gm;
  unsigned smask = 0;                  // feet whose leg block K_e is (numerically) rank deficient
  {
    const int f = (lane < 24) ? lane / 6 : 0, c = (lane < 24) ? lane - 6 * f : 0;
    int d0 = legd[0], d1 = legd[1], d2 = legd[2], rs = rowstart[0];
#pragma unroll
    for (int t = 1; t < 4; ++t) { const bool m = f == t; d0 = m ? legd[3 * t] : d0; d1 = m ? legd[3 * t + 1] : d1; d2 = m ? legd[3 * t + 2] : d2; rs = m ? rowstart[t] : rs; }
    const double* r0 = Co + rs * LDJ; const double* r1 = r0 + LDJ; const double* r2 = r1 + LDJ;
    const double k00 = r0[d0], k01 = r0[d1], k02 = r0[d2], k10 = r1[d0], k11 = r1[d1], k12 = r1[d2],
                 k20 = r2[d0], k21 = r2[d1], k22 = r2[d2];
    const double b0 = r0[c], b1 = r1[c], b2 = r2[c];
    const double a00 = k11 * k22 - k12 * k21, a01 = k02 * k21 - k01 * k22, a02 = k01 * k12 - k02 * k11;
    const double a10 = k12 * k20 - k10 * k22, a11 = k00 * k22 - k02 * k20, a12 = k02 * k10 - k00 * k12;
    const double a20 = k10 * k21 - k11 * k20, a21 = k01 * k20 - k00 * k21, a22 = k00 * k11 - k01 * k10;
    const double det = k00 * a00 + k01 * a10 + k02 * a20;
    const double sc = fabs(k00) + fabs(k01) + fabs(k02) + fabs(k10) + fabs(k11) + fabs(k12) + fabs(k20) + fabs(k21) + fabs(k22);
    const bool live = lane < 6 * nelim;
    const unsigned long long sb = __ballot(live && c == 0 && !(fabs(det) > A.sing_tol * sc * sc * sc));   // lanes 0, 6, 12, 18
    smask = (unsigned)((sb & 1ull) | ((sb >> 5) & 2ull) | ((sb >> 10) & 4ull) | ((sb >> 15) & 8ull));
    const double id = -1.0 / det;
    if (lane < 24) {
      Gm[(3 * f + 0) * GS + c] = live ? id * (a00 * b0 + a01 * b1 + a02 * b2) : 0.0;
      Gm[(3 * f + 1) * GS + c] = live ? id * (a10 * b0 + a11 * b1 + a12 * b2) : 0.0;
      Gm[(3 * f + 2) * GS + c] = live ? id * (a20 * b0 + a21 * b1 + a22 * b2) : 0.0;
    }
  }
  // ---- PIVOTED ELIMINATION of a rank-deficient leg block (rare; uniform branch). K_e P = Q R by column-pivoted Gram-Schmidt
  // (third direction = q0 x q1, so nothing is divided by the small pivot): with z = P'q̇_leg the contact rows read
  // Q'B q̇_base + R z = 0. The first two are solved for z0, z1 as before; the third, (Q'B)_2 q̇_base + r22 z2 = 0, is KEPT as an
  // equality row of the reduced QP and z2 — the velocity of the leg DoF pivoted last — stays an unknown of its own (reduced
  // variable n_red + j, column 6 + j of G). Exact, and as well conditioned as the rank-2 part of K_e; nothing is deferred unless a
  // block has rank < 2. Per flagged foot f, lane f does the 3 x 3 work; E (the kept equality row) and the pivot index go to the
  // dead tail of S.in.
  int nsing = 0;
  double* const Em = S.in + 96;        // [4][8]: 6 base coefficients, r22, index l of the leg DoF kept as unknown
  if (smask) {
    nsing = __popc(smask);
    bool bad_rank = A.dbg_force_defer != 0;
    if (A.pivot_count && lane == 0 && !bad_rank) atomicAdd(A.pivot_count, 1);
    if (lane < 48) Gm[(lane >> 2) * GS + 6 + (lane & 3)] = 0.0;     // extra columns of every row
    WSYNC();
    if (lane < 4 && ((smask >> lane) & 1u)) {
      const int f = lane, j = __popc(smask & ((1u << f) - 1u));
      int rs = rowstart[0], d0 = legd[0], d1 = legd[1], d2 = legd[2];
#pragma unroll
      for (int t = 1; t < 4; ++t) { const bool m = f == t; d0 = m ? legd[3 * t] : d0; d1 = m ? legd[3 * t + 1] : d1; d2 = m ? legd[3 * t + 2] : d2; rs = m ? rowstart[t] : rs; }
      const double* r0 = Co + rs * LDJ; const double* r1 = r0 + LDJ; const double* r2 = r1 + LDJ;
      // columns of K (as 3-vectors)
      double ca[3] = {r0[d0], r1[d0], r2[d0]}, cb[3] = {r0[d1], r1[d1], r2[d1]}, cc[3] = {r0[d2], r1[d2], r2[d2]};
      const double na = ca[0] * ca[0] + ca[1] * ca[1] + ca[2] * ca[2], nb = cb[0] * cb[0] + cb[1] * cb[1] + cb[2] * cb[2],
                   nc = cc[0] * cc[0] + cc[1] * cc[1] + cc[2] * cc[2];
      // first pivot: the longest column -> (u, then v, w the other two in index order)
      const int p0 = (na >= nb && na >= nc) ? 0 : ((nb >= nc) ? 1 : 2);
      double u[3], v[3], w[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        u[i] = (p0 == 0) ? ca[i] : (p0 == 1) ? cb[i] : cc[i];
        v[i] = (p0 == 0) ? cb[i] : ca[i];
        w[i] = (p0 == 2) ? cb[i] : cc[i];
      }
      const int iv = (p0 == 0) ? 1 : 0, iw = (p0 == 2) ? 1 : 2;
      const double r00 = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
      const double q0[3] = {u[0] / r00, u[1] / r00, u[2] / r00};
      const double rv = q0[0] * v[0] + q0[1] * v[1] + q0[2] * v[2], rw = q0[0] * w[0] + q0[1] * w[1] + q0[2] * w[2];
#pragma unroll
      for (int i = 0; i < 3; ++i) { v[i] = fma(-rv, q0[i], v[i]); w[i] = fma(-rw, q0[i], w[i]); }
      const double nv2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], nw2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
      const bool sw = nw2 > nv2;                       // second pivot: the longer remainder
      const int p1 = sw ? iw : iv, p2 = sw ? iv : iw;
      const double r01 = sw ? rw : rv, r02 = sw ? rv : rw;
      double s1[3], s2[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) { s1[i] = sw ? w[i] : v[i]; s2[i] = sw ? v[i] : w[i]; }
      const double r11 = sqrt(sw ? nw2 : nv2);
      const double q1[3] = {s1[0] / r11, s1[1] / r11, s1[2] / r11};
      const double r12 = q1[0] * s2[0] + q1[1] * s2[1] + q1[2] * s2[2];
      double q2[3];
      cross3(q0, q1, q2);
      const double r22 = q2[0] * s2[0] + q2[1] * s2[1] + q2[2] * s2[2];
      bad_rank = bad_rank || !(r11 > 1e-9 * r00) || !(r00 > 0.0);    // rank < 2 (or NaN): nothing sensible to eliminate
      // Q'B, then back substitution
      const int l0 = 3 * f + p0, l1 = 3 * f + p1, l2 = 3 * f + p2;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const double bx = r0[c], by = r1[c], bz = r2[c];
        const double t0 = q0[0] * bx + q0[1] * by + q0[2] * bz, t1 = q1[0] * bx + q1[1] * by + q1[2] * bz,
                     t2 = q2[0] * bx + q2[1] * by + q2[2] * bz;
        const double g1 = -t1 / r11;
        Gm[l1 * GS + c] = g1;
        Gm[l0 * GS + c] = -(t0 + r01 * g1) / r00;
        Gm[l2 * GS + c] = 0.0;
        Em[8 * f + c] = t2;
      }
      const double g1x = -r12 / r11;
      Gm[l1 * GS + 6 + j] = g1x;
      Gm[l0 * GS + 6 + j] = -(r01 * g1x + r02) / r00;
      Gm[l2 * GS + 6 + j] = 1.0;
      Em[8 * f + 6] = r22;
      Em[8 * f + 7] = (double)l2;
    }
    if (__ballot(bad_rank) || n_red + nsing > NR) {   // left to the general kernel's second pass (compact list)
      if (lane == 0) {
        A.out.status[b] = WBC_QP_DEFERRED;
        const int slot = atomicAdd(A.defer, 1);
        if (slot < A.B) A.defer[1 + slot] = b;   // (a stale count — a failed second-pass launch, one handle on two streams — must not write past the list)
      }
      WSYNC();
      return;
    }
  }
  WSYNC();
  STAMP(ts, T_P1);
  // per-lane views of the extra unknowns (all -1 / 0 without pivoted feet): ex_f = foot whose kept leg velocity is reduced
  // variable `lane`, ex_l its leg index, ex_d its DoF; my_x = reduced position of DoF `lane` if it is such a kept velocity
  const int n_eff = n_red + nsing;
  int ex_f = -1, ex_l = -1, my_x = -1;
  if (nsing) {
    int cnt = 0;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      if ((smask >> f) & 1u) {                     // uniform
        const int l2 = (int)Em[8 * f + 7];
        if (lane == n_red + cnt) { ex_f = f; ex_l = l2; }
        if (my_l == l2) my_x = n_red + cnt;
        ++cnt;
      }
    }
  }
  int ex_d = 0;
#pragma unroll
  for (int l = 0; l < 12; ++l) ex_d = (ex_l == l) ? legd[l] : ex_d;
  const bool is_ex = ex_f >= 0;
  const int fjb = is_ex ? ex_d : fj;   // DoF whose velocity bound / carried working-set bit belongs to reduced variable `lane`
  double gcol[12];
#pragma unroll
  for (int l = 0; l < 12; ++l) gcol[l] = (lane < 6) ? Gm[l * GS + lane] : (is_ex ? Gm[l * GS + 6 + (lane - n_red)] : 0.0);   // rows >= nl are zero
  // g' = Z'g
  double g_red = is_ex ? 0.0 : S.npv[fj];
#pragma unroll
  for (int l = 0; l < 12; ++l) g_red = fma(gcol[l], S.npv[legd[l]], g_red);
  if (lane >= n_eff) g_red = 0.0;
  STAMP(ts, T_P2);
  // ---- C' = C Z for the rows that stay, then the eliminated legs' bounds as rows G_l (for a pivoted foot the row of its kept
  // leg velocity holds the kept contact equality instead; that velocity's own bounds are variable bounds now)
  double* const Cm = S.RC;
  double nclb = 0.0, ncub = 0.0;
  int i2 = 0, my_orig = -1;            // my_orig: original index of the kept row that becomes reduced row `lane`
#pragma unroll 1
  for (int i = 0; i < p; ++i) {
    if ((elimrows >> i) & 1u) continue;
    double v = (lane < n_red) ? Co[i * LDJ + fj] : 0.0;
    if ((legrows >> i) & 1u) {
#pragma unroll
      for (int l = 0; l < 12; ++l) v = fma(gcol[l], Co[i * LDJ + legd[l]], v);
    }
    if (lane < CSC) Cm[i2 * CSC + lane] = v;
    const double bl = rdl(clb, i), bu = rdl(cub, i);
    if (lane == i2) { nclb = bl; ncub = bu; my_orig = i; }
    ++i2;
  }
  if (c_use_bounds) {
#pragma unroll
    for (int l = 0; l < 12; ++l) {
      if (l < nl) { if (lane < CSC) Cm[(i2 + l) * CSC + lane] = gcol[l]; }
    }
    if (lane >= i2 && lane < i2 + nl) { nclb = S.xv[my_legd]; ncub = S.yv[my_legd]; }
  }
  if (nsing) {
    if (!c_use_bounds) {               // no leg-bound rows to take over: clear the slots the kept equalities go into
#pragma unroll
      for (int l = 0; l < 12; ++l) { if (l < nl && lane < CSC) Cm[(i2 + l) * CSC + lane] = 0.0; }
      if (lane >= i2 && lane < i2 + nl) { nclb = -1e30; ncub = 1e30; }
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      if ((smask >> f) & 1u) {                     // uniform
        const int l2 = (int)Em[8 * f + 7];
        const double ev = (lane < 6) ? Em[8 * f + lane] : ((ex_f == f) ? Em[8 * f + 6] : 0.0);
        if (lane < CSC) Cm[(i2 + l2) * CSC + lane] = ev;
        if (lane == i2 + l2) { nclb = 0.0; ncub = 0.0; }
      }
    }
  }
  if (c_use_bounds || nsing) i2 += nl;
  const double lb_red = (lane < n_eff) ? S.xv[fjb] : 0.0, ub_red = (lane < n_eff) ? S.yv[fjb] : 0.0;
  STAMP(ts, T_P3);
  // ---- H' += d^2 G'G on the base block and the extra unknowns (H_ll = d^2 I, H_lf = 0: DevPlan.enabled)
  {
    const double d2 = dpost * dpost;
    double gg[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int l = 0; l < 12; ++l) {
#pragma unroll
      for (int c = 0; c < 6; ++c) gg[c] = fma(gcol[l], Gm[l * GS + c], gg[c]);
    }
    if (lane < 6 || is_ex) {
#pragma unroll
      for (int c = 0; c < 6; c += 2) {
        const double2a h2 = lane < 6 ? lds2(S.RA + lane * LDJ + c) : double2a{0.0, 0.0};     // (an extra unknown's row starts empty)
        sts2(S.RA + lane * LDJ + c, fma(d2, gg[c], h2.x), fma(d2, gg[c + 1], h2.y));
      }
    }
    if (nsing) {                                    // columns of the extra unknowns
      double gx[4] = {0, 0, 0, 0};
#pragma unroll
      for (int l = 0; l < 12; ++l) {
#pragma unroll
        for (int j = 0; j < 4; ++j) gx[j] = fma(gcol[l], Gm[l * GS + 6 + j], gx[j]);
      }
      if (lane < NR) {   // (this also replaces the identity padding's 1.0 on the extra unknowns' diagonal: the padding starts at n_eff)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j < nsing) S.RA[lane * LDJ + n_red + j] = (lane < 6 || is_ex) ? d2 * gx[j] : 0.0;
        }
      }
    }
  }
  WSYNC();
  STAMP(ts, T_PRE);
  DBG_STOP(5, g_red + lb_red + ub_red + nclb + ncub + S.RA[(lane & 15) * LDJ + 1] + Cm[(lane & 15) * CSC + 1]);
  // warm start: the carried working set (full-problem indexing, KernelArgs.ws_in) seen from the reduced problem — reduced
  // variable k is DoF Fd[k] (or the kept leg DoF of a pivoted foot); reduced row r is kept row my_orig or, from p_keep on, the
  // velocity bound of an eliminated leg DoF
  int sd_b = 0, sd_r = 0;
  if (WARM) {
    if (lane < n_eff) sd_b = (int)(((ws0 >> fjb) & 1ull) | (((ws0 >> (32 + fjb)) & 1ull) << 1));
    if (my_orig >= 0) sd_r = (int)(((ws1 >> my_orig) & 1ull) | (((ws1 >> (32 + my_orig)) & 1ull) << 1));
    else if (c_use_bounds && lane >= p_keep && lane < p_keep + nl) sd_r = (int)(((ws0 >> my_legd) & 1ull) | (((ws0 >> (32 + my_legd)) & 1ull) << 1));
  }
  QpResult res;
  if (n_eff <= 12) res = qp_core<12, SmemC, CSC, WARM>(S, g_red, lb_red, ub_red, nclb, ncub, n_eff, i2, lane, ts, DBG_STOP_ARG,
                                                       sd_b == 3 ? 0 : sd_b, sd_r == 3 ? 0 : sd_r);   // (no pivoted foot: n' = 11 / 10)
  else res = qp_core<NR, SmemC, CSC, WARM>(S, g_red, lb_red, ub_red, nclb, ncub, n_eff, i2, lane, ts, DBG_STOP_ARG,
                                           sd_b == 3 ? 0 : sd_b, sd_r == 3 ? 0 : sd_r);
  res.iters += nl - nsing + P.nlock;   // the eliminated equalities and the locked DoF, so that `iters` keeps its meaning
  // ---- x = Z y
  WSYNC();
  if (lane < 32) { S.xv[lane] = (lane < n_eff) ? res.x : 0.0; if (WARM) { S.lv[lane] = (double)res.ws_b; S.dinv[lane] = (double)res.ws_r; } }
  WSYNC();
  if (WARM && A.ws_out) {   // the final working set back in full-problem indexing: lane d = DoF d, lane i = original constraint row i
    int cb = 0, cr = 0;
    if (my_pos >= 0) cb = (int)S.lv[my_pos];
    else if (my_x >= 0) cb = (int)S.lv[my_x & 31];
    else if (my_l >= 0 && c_use_bounds) cb = (int)S.dinv[(p_keep + my_l) & 31];
    if (lane < p && !((elimrows >> lane) & 1u)) cr = (int)S.dinv[__popc(~elimrows & ((1u << lane) - 1u)) & 31];
    const unsigned long long o0 = (__ballot(cb == 1) & 0xFFFFFFFFull) | (__ballot(cb == 2) << 32);
    const unsigned long long o1 = (__ballot(cr == 1) & 0xFFFFFFFFull) | (__ballot(cr == 2) << 32);
    if (lane == 0) { A.ws_out[2 * (size_t)b] = o0; A.ws_out[2 * (size_t)b + 1] = o1; }
  }
  double x = 0.0;
  if (my_pos >= 0) x = S.xv[my_pos];
  else if (my_l >= 0) {
#pragma unroll
    for (int c = 0; c < 6; ++c) x = fma(Gm[my_l * GS + c], S.xv[c], x);
    if (nsing) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { if (j < nsing) x = fma(Gm[my_l * GS + 6 + j], S.xv[n_red + j], x); }
    }
  }
  if (lane >= nv) x = 0.0;
  if (A.out.qdot && lane < NV) A.out.qdot[(size_t)b * NV + lane] = x;
  if (lane == 0) {
    A.out.status[b] = res.status;
    if (A.out.iters) A.out.iters[b] = res.iters;
  }
  // ---- jointVelocitiestoConfig (Robot_Wrapper4.py:440-441)
  if (A.out.q_next) {
    double* qn = A.out.q_next + (size_t)b * NQ;
    const double v = x * dt;
    WSYNC();
    if (lane < 32) S.xv[lane] = (lane < nv) ? v : 0.0;
    WSYNC();
    integrate_ff(S, lane, qn);
    if (lane >= 6 && lane < nv) qn[lc.col_q] = qv[lc.col_q] + v;
    if (lane >= nq && lane < NQ) qn[lane] = 0.0;
    WSYNC();
  }
#ifdef WBC_PROFILE
  STAMP(ts, T_END);
  if (A.prof && lane == 0 && res.status == WBC_QP_OPTIMAL) {   // same slots as process_instance ([3] includes the presolve)
    for (int i = 1; i < T_N; ++i) atomicAdd(A.prof + i, ts[i] - ts[i - 1]);
    atomicAdd(A.prof + 0, 1ull);
    atomicAdd(A.prof + 8, (unsigned long long)res.iters);
    atomicAdd(A.prof + 9, ts[T_A1] - ts[T_FK]); atomicAdd(A.prof + 10, ts[T_A2] - ts[T_A1]);
    atomicAdd(A.prof + 11, ts[T_A3] - ts[T_A2]); atomicAdd(A.prof + 12, ts[T_ASM] - ts[T_A3]);
    atomicAdd(A.prof + 13, ts[T_PRE] - ts[T_ASM]); atomicAdd(A.prof + 14, 1ull);
    atomicAdd(A.prof + 16, ts[T_P1] - ts[T_ASM]); atomicAdd(A.prof + 17, ts[T_P2] - ts[T_P1]);
    atomicAdd(A.prof + 18, ts[T_P3] - ts[T_P2]); atomicAdd(A.prof + 19, ts[T_PRE] - ts[T_P3]);
  }
#endif
}

template <bool WARM>
__global__ void __launch_bounds__(64, 3) wbc_tick_sim3_kernel(const KernelArgs A, const DevModel* __restrict__ models,
                                                              const WbcConfig* __restrict__ cfgs, const DevPlan* __restrict__ plans) {
  __shared__ SmemC S;
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  if (lane < 48) S.cl[lane] = 0.0;
  const bool has2 = A.in.trunk_target || A.in.prev_trunk_target || A.in.trunk_ref_euler || A.in.trunk_prev_rot;
  const int mid = model_index(A.in.model_id, b, A.n_models);
  const InRegs cur = load_inputs(A.in, A.dbg_alias ? 0 : b, lane, has2, false);
  const LaneConst lc = load_lane_const(models[mid], cfgs[mid], lane);
  Hdr H = load_hdr(models[mid]);                 // one batch of scalar loads, waited for once
  asm volatile("" : "+s"(H.nq), "+s"(H.nv), "+s"(H.nj), "+s"(H.maxdepth), "+s"(H.nframes), "+s"(H.trunk_joint));
  // the carried working set (warm start): two uniform words, fetched with the other inputs
  const unsigned long long ws0 = (WARM && A.ws_in) ? A.ws_in[2 * (size_t)b] : 0ull, ws1 = (WARM && A.ws_in) ? A.ws_in[2 * (size_t)b + 1] : 0ull;
  stage_inputs(S, cur, lane, has2, false);
  WSYNC();
  process_sim3<WARM>(S, A, models[mid], cfgs[mid], plans[mid], H, lc, cur, b, lane, ws0, ws1);
}

// Second pass after wbc_tick_sim3_kernel: the instances it deferred (a stance-leg block it could not eliminate) are redone on
// the general path. The sim3 kernel appended them to a compact list (A.defer: count, then instance indices, in arrival
// order); workgroup i takes entries i, i + gridDim.x, ... — with at most gridDim.x deferred instances (the usual handful)
// every one has a workgroup of its own, and a batch that defers everything is spread over the whole chip instead of being
// walked 64 instances per wave. Every wave reaches the loop exit (i >= count).
__global__ void __launch_bounds__(64, 2) wbc_tick_deferred_kernel(const KernelArgs A, const DevModel* __restrict__ models,
                                                                  const WbcConfig* __restrict__ cfgs, const DevPlan* __restrict__ plans) {
  __shared__ Smem S;
  const int lane0 = threadIdx.x;
  const int count_raw = __builtin_amdgcn_readfirstlane(A.defer[0]);
  const int count = count_raw > A.B ? A.B : count_raw;
  // The list resets itself (no memset launch per tick — at small batches that dispatch was a tenth of the tick): an empty list is
  // already zero; otherwise the workgroups that had work count themselves out and the last one to finish — every other one has read
  // the count by then, and a workgroup that has not started yet has index >= count whatever it reads — clears it. defer_aux: [1] done
  // counter, [2] the count for the "deferred_last" statistic.
  if (count == 0) { if (blockIdx.x == 0 && lane0 == 0) A.defer_aux[2] = 0; return; }
  if ((int)blockIdx.x >= count) return;
  const bool has2 = A.in.trunk_target || A.in.prev_trunk_target || A.in.trunk_ref_euler || A.in.trunk_prev_rot ||
                    A.in.com_target || A.in.com_target_vel;
  const bool has3 = A.in.ee_ref_rot != nullptr;
#pragma unroll 1
  for (int i = blockIdx.x; i < count; i += gridDim.x) {
    int b = __builtin_amdgcn_readfirstlane(A.defer[1 + i]);
    b = b < 0 ? 0 : (b >= A.B ? A.B - 1 : b);
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    S.cl[lane] = 0.0;
    const int mid = model_index(A.in.model_id, b, A.n_models);
    const InRegs cur = load_inputs(A.in, b, lane, has2, has3);
    const LaneConst lc = load_lane_const(models[mid], cfgs[mid], lane);
    stage_inputs(S, cur, lane, has2, has3);
    WSYNC();
    process_instance<MODE_TICK, true>(S, A, models[mid], cfgs[mid], plans[mid], lc, cur, b, lane, 0ull);
    WSYNC();
  }
  if (lane0 == 0) {
    __threadfence();
    const int nblk = count < (int)gridDim.x ? count : (int)gridDim.x;
    if (atomicAdd(A.defer_aux + 1, 1) == nblk - 1) { A.defer_aux[2] = count_raw; A.defer_aux[1] = 0; A.defer[0] = 0; }
  }
}

// Stand-alone QP (QP_Wrapper.QP.solveQP): H, g (or A, b) and constraints straight from HBM.
// NM: compiled size of the dual active-set core (launch_qp picks the smallest of 12 / 16 / 24 / 26 that holds n: its sweeps cost ~NM^2)
// WARM: working sets in / out (QP.solveQPHotstart, QP_Wrapper.py:55-73): [B][2] words in the problem's own indexing — word 0: bit i / 32 + i =
// variable i at its lower / upper bound, word 1: constraint row i
template <int NM, bool WARM = false>
__global__ void __launch_bounds__(64, 2) wbc_qp_kernel(const QpArgs A) {
  __shared__ Smem S;
  S.cl[threadIdx.x] = 0.0;
#pragma unroll 1
  for (int b = blockIdx.x; b < A.B; b += gridDim.x) {
    int lane = threadIdx.x, n = A.n, p = A.p, m = A.m;
    asm volatile("" : "+v"(lane), "+s"(n), "+s"(p), "+s"(m));   // no LICM of lane/n-derived masks
    const int li = li_clamp(lane);
    double g = 0.0;
    if (m > 0) {
      // H = A'A, g = -A'b (QP_Wrapper.py:17-18)
      const double* Ab = A.A + (size_t)b * m * n;
      const double* bb = A.bvec + (size_t)b * m;
      if (A.use_mfma) {
        // matrix-core path: column 26 of the padded operand carries b, so A'b comes out of the same MFMAs
        jtj_mfma(S, lane, m, [&](int r, int c) -> double {
          if (r >= m) return 0.0;
          if (c < n) return Ab[(size_t)r * n + c];
          return (c == NV) ? bb[r] : 0.0;
        });
        g = (lane < n) ? -S.npv[li] : 0.0;
        if (n < NV) {                                   // padded rows / columns of H are exactly zero off the diagonal
          WSYNC();
        }
      } else {
        // vector path: A is staged by DoF (At[dof][row]) in chunks of <= 48 rows; H accumulates in RA
        double* const At = S.RB;
        constexpr int CH = 48, MT = 50;
        if (lane < NV) for (int k = 0; k < NV; k += 2) sts2(S.RA + lane * LDJ + k, 0.0, 0.0);
#pragma unroll 1
        for (int r0 = 0; r0 < m; r0 += CH) {
          const int mc = (m - r0 < CH) ? m - r0 : CH;
#pragma unroll 1
          for (int r = 0; r < CH; ++r) {
            const double a = (r < mc && lane < n) ? Ab[(size_t)(r0 + r) * n + lane] : 0.0;
            if (lane < NV) At[lane * MT + r] = a;
            if (r < mc) g = fma(-a, bb[r0 + r], g);
          }
          WSYNC();
#pragma unroll 1
          for (int i = 0; i < n; ++i) {
            double s = S.RA[li * LDJ + i];
#pragma unroll 2
            for (int r = 0; r < CH; r += 2) {
              const double2a x2 = lds2(At + i * MT + r); const double2a y2 = lds2(At + li * MT + r);
              s = fma(x2.x, y2.x, fma(x2.y, y2.y, s));
            }
            if (lane < NV) S.RA[lane * LDJ + i] = s;
          }
          WSYNC();
        }
      }
      if (lane >= n && lane < NV) S.RA[lane * LDJ + lane] = 1.0;   // padded DoF
      WSYNC();
      if (A.H_out && lane < n) { double* o = A.H_out + (size_t)b * n * n + (size_t)lane * n; for (int k = 0; k < n; ++k) o[k] = S.RA[lane * LDJ + k]; }
      if (A.g_out && lane < n) A.g_out[(size_t)b * n + lane] = g;
    } else {
      const double* Hb = A.H + (size_t)b * n * n;
#pragma unroll 1
      for (int idx = lane; idx < NV * NV; idx += 64) {
        const int r = idx / NV, c = idx - r * NV;
        S.RA[r * LDJ + c] = (r < n && c < n) ? Hb[(size_t)r * n + c] : ((r == c) ? 1.0 : 0.0);
      }
      g = (lane < n) ? A.g[(size_t)b * n + lane] : 0.0;
    }
#pragma unroll 1
    for (int r = 0; r < p; ++r)
      if (lane < NV) S.RC[r * LDJ + lane] = (lane < n) ? A.C[((size_t)b * p + r) * n + lane] : 0.0;
    const double lb = (lane < n) ? (A.lb ? A.lb[(size_t)b * n + lane] : -1e30) : 0.0;
    const double ub = (lane < n) ? (A.ub ? A.ub[(size_t)b * n + lane] : 1e30) : 0.0;
    const double clb = (lane < p) ? A.Clb[(size_t)b * p + lane] : 0.0;
    const double cub = (lane < p) ? A.Cub[(size_t)b * p + lane] : 0.0;
    WSYNC();
    unsigned long long ts[T_NN];
    (void)ts;
    int sb = 0, sr = 0;
    if (WARM && A.ws_in) {
      const unsigned long long w0 = A.ws_in[2 * (size_t)b], w1 = A.ws_in[2 * (size_t)b + 1];
      sb = (lane < 32) ? (int)(((w0 >> lane) & 1ull) | (((w0 >> (32 + lane)) & 1ull) << 1)) : 0;
      sr = (lane < 32) ? (int)(((w1 >> lane) & 1ull) | (((w1 >> (32 + lane)) & 1ull) << 1)) : 0;
    }
    const QpResult res = qp_core<NM, Smem, LDJ, WARM>(S, g, lb, ub, clb, cub, n, p, lane, ts, 0, sb == 3 ? 0 : sb, sr == 3 ? 0 : sr);
    if (WARM && A.ws_out) {
      const unsigned long long o0 = (__ballot(res.ws_b == 1) & 0xFFFFFFFFull) | (__ballot(res.ws_b == 2) << 32);
      const unsigned long long o1 = (__ballot(res.ws_r == 1) & 0xFFFFFFFFull) | (__ballot(res.ws_r == 2) << 32);
      if (lane == 0) { A.ws_out[2 * (size_t)b] = o0; A.ws_out[2 * (size_t)b + 1] = o1; }
    }
    if (lane < n) A.x[(size_t)b * n + lane] = res.x;
    if (lane == 0) {
      if (A.status) A.status[b] = res.status;
      if (A.iters) A.iters[b] = res.iters;
    }
    WSYNC();
  }
}

// pin.integrate for a batch (Robot_Wrapper4.py:440-441): q_next = q (+) v * dt
__global__ void __launch_bounds__(64) wbc_integrate_kernel(const IntegrateArgs A) {
  __shared__ Smem S;
#pragma unroll 1
  for (int b = blockIdx.x; b < A.B; b += gridDim.x) {
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));
    const DevModel& M = A.models[model_index(A.model_id, b, A.n_models)];
    const int nv = M.nv, nq = M.nq;
    if (lane < 32) S.in[IN_Q + lane] = (lane < nq) ? A.q[(size_t)b * NQ + lane] : 0.0;
    const double v = (lane < nv) ? A.v[(size_t)b * NV + lane] * A.dt : 0.0;
    if (lane < 32) S.xv[lane] = v;
    WSYNC();
    double* qn = A.q_next + (size_t)b * NQ;
    integrate_ff(S, lane, qn);
    const int cq = M.col_q[lane & 31];
    if (lane >= 6 && lane < nv) qn[cq] = S.in[IN_Q + cq] + v;
    if (lane >= nq && lane < NQ) qn[lane] = 0.0;
    WSYNC();
  }
}

// ------------------------------------------------------------------------------------------------
// qpJointb "MANI" / "HYBRID" (Robot_Wrapper4.py:1220-1260): u_i = (f(q + d e) - f(q - d e)) / (2 d), f = sqrt(det(J J'))
// of pin.getJointJacobian(joint_id, LOCAL_WORLD_ALIGNED); one instance per wave, the 2 x (6 or 26) perturbed
// configurations are evaluated one after the other (each is a full FK: lane j = joint j, then lane k = column k).
// literal (cfg.posture_literal): the reference's index arithmetic and accumulating perturbations.
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) PSmem {
  double oMi[24 * 12];
  double q[32];
  double Jm[32 * 6];     // column k of the 6 x nv joint Jacobian at Jm[6 k ..]
  double G[36];
};
// det of the symmetric positive semi-definite G = J J' by elimination without pivoting (numpy's det pivots; for an SPD
// matrix both are backward stable and agree to rounding). G is wave-uniform in LDS.
__device__ __forceinline__ double det6_spd(const double* G) {
  double m[6][6];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) m[a][b] = G[6 * a + b];
  double det = 1.0;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const double piv = m[c][c];
    det *= piv;
    const double ip = (piv > 0.0) ? 1.0 / piv : 0.0;
#pragma unroll
    for (int r = c + 1; r < 6; ++r) {
      const double f = m[r][c] * ip;
#pragma unroll
      for (int k = c + 1; k < 6; ++k) m[r][k] = fma(-f, m[c][k], m[r][k]);
    }
  }
  return det > 0.0 ? det : 0.0;
}
__device__ __forceinline__ double manipulability(PSmem& P, const DevModel& M, const LaneConst& lc, const int joint_id,
                                                 const int lane) {
  const int nv = M.nv;
  fk_levels(P.oMi, P.q, load_hdr(M), lc, lane);
  double lin[3], ang[3];
  jac_column(P.oMi, lc, lane, nv, lin, ang);
  const double* Pj = P.oMi + 12 * joint_id;
  const double pj[3] = {Pj[9], Pj[10], Pj[11]};
  const bool sup = (lane < nv) && ((lc.subtree >> joint_id) & 1u);   // column's joint is joint_id or one of its ancestors
  double wxp[3];
  cross3(ang, pj, wxp);
  if (lane < 32) {
#pragma unroll
    for (int r = 0; r < 3; ++r) { P.Jm[6 * lane + r] = sup ? lin[r] + wxp[r] : 0.0; P.Jm[6 * lane + 3 + r] = sup ? ang[r] : 0.0; }
  }
  WSYNC();
  if (lane < 36) {
    const int a = lane / 6, b = lane - 6 * a;
    double s = 0.0;
#pragma unroll 2
    for (int k = 0; k < NV; ++k) s = fma(P.Jm[6 * k + a], P.Jm[6 * k + b], s);
    P.G[lane] = s;
  }
  WSYNC();
  const double f = sqrt(det6_spd(P.G));
  WSYNC();
  return f;
}

__global__ void __launch_bounds__(64) wbc_posture_kernel(const PostureArgs A, const DevModel* __restrict__ models,
                                                         const WbcConfig* __restrict__ cfgs) {
  __shared__ PSmem P;
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int mid = model_index(A.model_id, b, A.n_models);
  const DevModel& M = models[mid];
  const WbcConfig& cfg = cfgs[mid];
  const LaneConst lc = load_lane_const(M, cfg, lane);
  const int nv = M.nv, nq = M.nq;
  const double q0 = (lane < nq) ? A.q[(size_t)b * NQ + lane] : 0.0;
  if (lane < 32) P.q[lane] = q0;
  WSYNC();
  const int mode = cfg.task_joint, literal = cfg.posture_literal;
  const double dq = 0.0002;
  double u = 0.0;
  if ((mode == WBC_JOINT_PREV || mode == WBC_JOINT_HYBRID) && lane < nv) u = P.q[lane < 6 ? lane : lane + 1];   // np.delete(q, 6)
  if (mode == WBC_JOINT_MANI || mode == WBC_JOINT_HYBRID) {
#pragma unroll 1
    for (int i = 0; i < nv; ++i) {
      int joint_id;
      if (mode == WBC_JOINT_MANI) joint_id = (i < 6) ? 1 : (literal ? i + 1 - 5 : i - 4);   // :1226-1229
      else { joint_id = i - 6; if (joint_id < cfg.arm_base_id) continue; }                 // :1250-1251
      if (joint_id >= M.njoints) continue;
      const int qi = literal ? i : ((i < 6) ? i : i + 1);                                   // q[i]: the VELOCITY index (:1231, :1252)
      const double keep = P.q[qi];
      WSYNC();
      // Which joint's angle is q[qi]? The LWA Jacobian of joint_id is built from the axes and origins of its PROPER
      // ancestors and its own origin/axis, none of which the FK derives from the angle of joint_id itself or of any joint
      // outside its ancestor chain (the axis column is an exact copy of the parent's, the origin does not involve the
      // angle). For such a perturbation f1 and f2 are computed from bit-identical inputs, so u_i = 0.5 (f1 - f2)/dq = 0
      // exactly — the twelve FK sweeps of the reference's HYBRID indices all fall in this class.
      const unsigned long long own = __ballot(lc.is_joint && lc.q_idx == qi && lane >= 2);
      const int jp = own ? ctz64(own) : 1;                                                 // qi < 7: the free-flyer (affects everything)
      const bool matters = (jp == 1) || (jp != joint_id && ((M.col_subtree[M.idx_v_of[jp]] >> joint_id) & 1u));
      if (lane == 0) P.q[qi] = keep + dq;
      WSYNC();
      double f1 = 0.0, f2 = 0.0;
      if (matters) f1 = manipulability(P, M, lc, joint_id, lane);
      if (lane == 0) P.q[qi] = (keep + dq) - (dq * 2);
      WSYNC();
      if (matters) f2 = manipulability(P, M, lc, joint_id, lane);
      if (lane == i) u = 0.5 * (f1 - f2) / dq;
      if (!literal) { if (lane == 0) P.q[qi] = keep; WSYNC(); }
    }
  }
  WSYNC();
  if (A.u && lane < NV) A.u[(size_t)b * NV + lane] = (lane < nv) ? u : 0.0;
  if (A.q_after && lane < NQ) A.q_after[(size_t)b * NQ + lane] = (lane < nq) ? P.q[lane] : 0.0;
}

// ------------------------------------------------------------------------------------------------
// qpJointb "MANI" / "HYBRID" (Robot_Wrapper4.py:1220-1260) with every finite-difference point on a LANE OF ITS OWN (round 3): one instance per
// wavefront, lane e = 2 k + side evaluates f = sqrt(det(J J')) of sweep k's joint at q + d e_i (side 0) or (q + d e_i) - 2 d e_i (side 1).
// The reference's loop is sequential only in appearance: in literal mode the perturbations accumulate, but the state sweep k
// sees is known up front — q0 with the entries of the earlier sweeps at (q + d) - 2 d (DevPlan.mp_prev) — so all 2 x mp_n (<= 52) points
// are independent. A lane walks the ancestor chain of its joint once in registers (no cross-lane traffic), forms the WORLD-frame Jacobian
// columns on the way (six of the free-flyer + one per chain joint; det(J J') is the same in every frame the columns may be expressed in)
// and accumulates them straight into G = J J' (21 entries), then det by the elimination of wbc_posture_kernel. sin / cos of every joint angle in its three possible states (q, q + d,
// (q + d) - 2 d) are computed once, one per lane, and shared through LDS. Sweeps that cannot change f (DevPlan: not in the list) are u = 0.
// wbc_posture_kernel (52 sequential whole-tree sweeps per instance) stays as the fallback and as the cross-check in the tests.
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) MPSmem {
  double q[32];
  double sc[24 * 6];      // joint j: sin, cos of q, of q + d, of (q + d) - 2 d
  double f[64];
  double uo[32];          // u of the swept DoF
};
__global__ void __launch_bounds__(64) wbc_posture_par_kernel(const PostureArgs A, const DevModel* __restrict__ models,
                                                             const WbcConfig* __restrict__ cfgs, const DevPlan* __restrict__ plans) {
  __shared__ MPSmem S;
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int mid = model_index(A.model_id, b, A.n_models);
  const DevModel& M = models[mid];
  const DevPlan& P = plans[mid];
  const int nv = M.nv, nq = M.nq, nj = M.njoints;
  const double dq = 0.0002;
  const double q0 = (lane < nq) ? A.q[(size_t)b * NQ + lane] : 0.0;
  if (lane < 32) S.q[lane] = q0;
  // this lane's evaluation
  const int k = lane >> 1, side = lane & 1;
  const bool on = k < P.mp_n;
  const int kk = on ? k : 0;
  const int my_i = P.mp_i[kk], my_qi = P.mp_qi[kk];
  const unsigned my_prev = P.mp_prev[kk];
  int chain[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) chain[c] = on ? P.mp_chain[kk][c] : -1;
  WSYNC();
  // sin / cos table: lane t < 3 (nj - 2): joint 2 + t / 3 in state t % 3
  {
    const int j = 2 + lane / 3, st = lane - 3 * (lane / 3);
    if (j < nj) {
      const int jt = M.jtype[j];
      if (jt >= WBC_JT_RX && jt <= WBC_JT_RZ) {
        const double a0 = S.q[M.idx_q[j]];
        const double a = (st == 0) ? a0 : ((st == 1) ? a0 + dq : (a0 + dq) - (dq * 2));
        const SinCos t = sincos_cw(a);
        S.sc[6 * j + 2 * st] = t.s; S.sc[6 * j + 2 * st + 1] = t.c;
      }
    }
  }
  WSYNC();
  // state of configuration entry e for this lane: 0 = q, 1 = q + d, 2 = (q + d) - 2 d
  auto state_of = [&](const int e) -> int { return (e == my_qi) ? (side ? 2 : 1) : (((my_prev >> e) & 1u) ? 2 : 0); };
  auto value_of = [&](const int e) -> double {
    const double a0 = S.q[e];
    const int st = state_of(e);
    return (st == 0) ? a0 : ((st == 1) ? a0 + dq : (a0 + dq) - (dq * 2));
  };
  double f = 0.0;
  {
    // the free-flyer: R from the (possibly perturbed, not renormalised) quaternion exactly as the FK does, p = xyz
    const double qq[4] = {value_of(3), value_of(4), value_of(5), value_of(6)};
    double R1[9];
    quat_to_R(qq, R1);                                  // row-major
    double G[21];
#pragma unroll
    for (int i = 0; i < 21; ++i) G[i] = 0.0;
    auto add_col = [&](const double* c) {               // G += c c' (upper triangle, row-major packed)
      int t = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int bb = a; bb < 6; ++bb) { G[t] = fma(c[a], c[bb], G[t]); ++t; }
    };
    // ONE walk down the chain with the columns expressed at the BASE origin ((p - p_base) x axis, axis): the reference's LOCAL_WORLD_ALIGNED
    // Jacobian at the joint's origin pJ is X J_base with X = [I, -[pJ - p_base]x; 0, I], det X = 1, so det(J J') — all that f is — does not
    // depend on where the columns are expressed, and the joint's origin need not be known before the columns are formed (the first version
    // walked the chain twice for it). Origins relative to the base: the base position drops out of the arithmetic altogether, so its three
    // sweeps give f1 == f2 bit for bit and u = 0 exactly, as the reference's (and the sequential kernel's) LOCAL_WORLD_ALIGNED form does.
    {
      double X[3] = {R1[0], R1[3], R1[6]}, Y[3] = {R1[1], R1[4], R1[7]}, Z[3] = {R1[2], R1[5], R1[8]};   // columns of the parent's rotation
      double p[3] = {0.0, 0.0, 0.0};                    // origins relative to the base
      // free-flyer columns at its own origin: linear DoF (R e_i, 0), angular DoF (0, R e_i)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double* ax = (i == 0) ? X : ((i == 1) ? Y : Z);
        const double cl[6] = {ax[0], ax[1], ax[2], 0.0, 0.0, 0.0};
        add_col(cl);
        const double ca[6] = {0.0, 0.0, 0.0, ax[0], ax[1], ax[2]};
        add_col(ca);
      }
#pragma unroll 1
      for (int c = 0; c < 8; ++c) {
        const int j = chain[c];
        if (j < 0) continue;
        const int a = M.ax0[j], jt = M.jtype[j];
        const bool rev = jt >= WBC_JT_RX && jt <= WBC_JT_RZ;
        const double t0 = M.tp[j][0], t1 = M.tp[j][1], t2 = M.tp[j][2];
        const int qe = M.idx_q[j];
        const int st = state_of(qe);
        const double sn = rev ? S.sc[6 * j + 2 * st] : 0.0, cs = rev ? S.sc[6 * j + 2 * st + 1] : 1.0;
        const double pris = rev ? 0.0 : value_of(qe);
        double Av[3], Bv[3], Cv[3];                     // the axis column of the parent's rotation and its cyclic successors
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          Av[rr] = (a == 0) ? X[rr] : ((a == 1) ? Y[rr] : Z[rr]);
          Bv[rr] = (a == 0) ? Y[rr] : ((a == 1) ? Z[rr] : X[rr]);
          Cv[rr] = (a == 0) ? Z[rr] : ((a == 1) ? X[rr] : Y[rr]);
        }
        double nB[3], nC[3];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          nB[rr] = cs * Bv[rr] + sn * Cv[rr];
          nC[rr] = cs * Cv[rr] - sn * Bv[rr];
          p[rr] = p[rr] + Av[rr] * (t0 + pris) + Bv[rr] * t1 + Cv[rr] * t2;
        }
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          const double na = Av[rr], nb = nB[rr], nc = nC[rr];
          X[rr] = (a == 0) ? na : ((a == 1) ? nc : nb);
          Y[rr] = (a == 0) ? nb : ((a == 1) ? na : nc);
          Z[rr] = (a == 0) ? nc : ((a == 1) ? nb : na);
        }
        double col[6];                                  // this joint's column: revolute (p x axis, axis), prismatic (axis, 0)
        if (rev) {
          double cr[3];
          cross3(p, Av, cr);
          col[0] = cr[0]; col[1] = cr[1]; col[2] = cr[2]; col[3] = Av[0]; col[4] = Av[1]; col[5] = Av[2];
        } else { col[0] = Av[0]; col[1] = Av[1]; col[2] = Av[2]; col[3] = 0.0; col[4] = 0.0; col[5] = 0.0; }
        add_col(col);
      }
    }
    // det of the symmetric positive semi-definite G by elimination without pivoting (det6_spd, on the packed upper triangle)
    double m[6][6];
    {
      int t = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int bb = a; bb < 6; ++bb) { m[a][bb] = G[t]; m[bb][a] = G[t]; ++t; }
    }
    double det = 1.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double piv = m[c][c];
      det *= piv;
      const double ip = (piv > 0.0) ? 1.0 / piv : 0.0;
#pragma unroll
      for (int rr = c + 1; rr < 6; ++rr) {
        const double ff = m[rr][c] * ip;
#pragma unroll
        for (int kx = c + 1; kx < 6; ++kx) m[rr][kx] = fma(-ff, m[c][kx], m[rr][kx]);
      }
    }
    f = sqrt(det > 0.0 ? det : 0.0);
  }
  S.f[lane] = on ? f : 0.0;
  WSYNC();
  // u by DoF: the sweep's central difference, the PREV value where the loop skips the DoF (HYBRID), else 0
  double u = 0.0;
  if (lane < nv && ((P.mp_prevmode >> lane) & 1u)) u = S.q[lane < 6 ? lane : lane + 1];
  if (on && side == 0) S.uo[my_i & 31] = 0.5 * (S.f[lane] - S.f[lane + 1]) / dq;   // lane 2 k holds f1, lane 2 k + 1 f2 of sweep k
  WSYNC();
  if (lane < nv) {
    bool mine = false;
#pragma unroll 1
    for (int t = 0; t < P.mp_n; ++t) mine |= (P.mp_i[t] == lane);
    if (mine) u = S.uo[lane];
  }
  if (A.u && lane < NV) A.u[(size_t)b * NV + lane] = (lane < nv) ? u : 0.0;
  if (A.q_after && lane < NQ) {
    const double a0 = S.q[lane & 31];
    A.q_after[(size_t)b * NQ + lane] = (lane < nq) ? (((P.mp_all >> lane) & 1u) ? (a0 + dq) - (dq * 2) : a0) : 0.0;
  }
}

// THREE instances per wavefront where no model of the batch has more than 21 sweeps (A1 + wx200 / px100 "MANI": 21 / 20): instance r = lane / 21,
// sweep k = lane % 21, and the lane evaluates BOTH sides of its central difference one after the other — 63 of 64 lanes busy where the kernel above
// keeps 42; the same arithmetic per evaluation (bit-identical u). Roles with more than 21 entries (configuration, sin / cos table, outputs) take
// two or three rounds of the instance's 21 lanes.
__global__ void __launch_bounds__(64) wbc_posture_par3_kernel(const PostureArgs A, const DevModel* __restrict__ models,
                                                              const WbcConfig* __restrict__ cfgs, const DevPlan* __restrict__ plans) {
  __shared__ MPSmem S3[3];
  const int lane0 = threadIdx.x;
  const bool grp = lane0 < 63;                             // (lane 63: no instance; it shadows instance 2's last lane and stores nothing)
  const int r = grp ? lane0 / 21 : 2, lane = grp ? lane0 - 21 * r : 20;
  MPSmem& S = S3[r];
  const int b_raw = 3 * (int)blockIdx.x + r;
  const bool valid = grp && b_raw < A.B;
  const int b = b_raw < A.B ? b_raw : A.B - 1;
  int mid = 0;
  if (A.model_id) { mid = A.model_id[b]; mid = mid < 0 ? 0 : (mid >= A.n_models ? A.n_models - 1 : mid); }
  const DevModel& M = models[mid];
  const DevPlan& P = plans[mid];
  const int nv = M.nv, nq = M.nq, nj = M.njoints;
  const double dq = 0.0002;
  if (grp) {
    S.q[lane] = (lane < nq) ? A.q[(size_t)b * NQ + lane] : 0.0;
    if (lane + 21 < 32) S.q[lane + 21] = (lane + 21 < nq) ? A.q[(size_t)b * NQ + lane + 21] : 0.0;
  }
  // this lane's sweep
  const int k = lane;
  const bool on = grp && k < P.mp_n;
  const int kk = on ? k : 0;
  const int my_i = P.mp_i[kk], my_qi = P.mp_qi[kk];
  const unsigned my_prev = P.mp_prev[kk];
  int chain[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) chain[c] = on ? P.mp_chain[kk][c] : -1;
  WSYNC();
  // sin / cos table: entry t < 3 (nj - 2): joint 2 + t / 3 in state t % 3; three rounds of 21 lanes
#pragma unroll 1
  for (int i = 0; i < 3; ++i) {
    const int tt = lane + 21 * i;
    const int j = 2 + tt / 3, st = tt - 3 * (tt / 3);
    if (grp && j < nj) {
      const int jt = M.jtype[j];
      if (jt >= WBC_JT_RX && jt <= WBC_JT_RZ) {
        const double a0 = S.q[M.idx_q[j]];
        const double a = (st == 0) ? a0 : ((st == 1) ? a0 + dq : (a0 + dq) - (dq * 2));
        const SinCos tsc = sincos_cw(a);
        S.sc[6 * j + 2 * st] = tsc.s; S.sc[6 * j + 2 * st + 1] = tsc.c;
      }
    }
  }
  WSYNC();
  double f1 = 0.0, f2 = 0.0;
#pragma unroll 1
  for (int side = 0; side < 2; ++side) {
  // state of configuration entry e for this lane: 0 = q, 1 = q + d, 2 = (q + d) - 2 d
  auto state_of = [&](const int e) -> int { return (e == my_qi) ? (side ? 2 : 1) : (((my_prev >> e) & 1u) ? 2 : 0); };
  auto value_of = [&](const int e) -> double {
    const double a0 = S.q[e];
    const int st = state_of(e);
    return (st == 0) ? a0 : ((st == 1) ? a0 + dq : (a0 + dq) - (dq * 2));
  };
  double f = 0.0;
  {
    // the free-flyer: R from the (possibly perturbed, not renormalised) quaternion exactly as the FK does, p = xyz
    const double qq[4] = {value_of(3), value_of(4), value_of(5), value_of(6)};
    double R1[9];
    quat_to_R(qq, R1);                                  // row-major
    double G[21];
#pragma unroll
    for (int i = 0; i < 21; ++i) G[i] = 0.0;
    auto add_col = [&](const double* c) {               // G += c c' (upper triangle, row-major packed)
      int t = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int bb = a; bb < 6; ++bb) { G[t] = fma(c[a], c[bb], G[t]); ++t; }
    };
    // ONE walk down the chain with the columns expressed at the BASE origin ((p - p_base) x axis, axis): the reference's LOCAL_WORLD_ALIGNED
    // Jacobian at the joint's origin pJ is X J_base with X = [I, -[pJ - p_base]x; 0, I], det X = 1, so det(J J') — all that f is — does not
    // depend on where the columns are expressed, and the joint's origin need not be known before the columns are formed (the first version
    // walked the chain twice for it). Origins relative to the base: the base position drops out of the arithmetic altogether, so its three
    // sweeps give f1 == f2 bit for bit and u = 0 exactly, as the reference's (and the sequential kernel's) LOCAL_WORLD_ALIGNED form does.
    {
      double X[3] = {R1[0], R1[3], R1[6]}, Y[3] = {R1[1], R1[4], R1[7]}, Z[3] = {R1[2], R1[5], R1[8]};   // columns of the parent's rotation
      double p[3] = {0.0, 0.0, 0.0};                    // origins relative to the base
      // free-flyer columns at its own origin: linear DoF (R e_i, 0), angular DoF (0, R e_i)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double* ax = (i == 0) ? X : ((i == 1) ? Y : Z);
        const double cl[6] = {ax[0], ax[1], ax[2], 0.0, 0.0, 0.0};
        add_col(cl);
        const double ca[6] = {0.0, 0.0, 0.0, ax[0], ax[1], ax[2]};
        add_col(ca);
      }
#pragma unroll 1
      for (int c = 0; c < 8; ++c) {
        const int j = chain[c];
        if (j < 0) continue;
        const int a = M.ax0[j], jt = M.jtype[j];
        const bool rev = jt >= WBC_JT_RX && jt <= WBC_JT_RZ;
        const double t0 = M.tp[j][0], t1 = M.tp[j][1], t2 = M.tp[j][2];
        const int qe = M.idx_q[j];
        const int st = state_of(qe);
        const double sn = rev ? S.sc[6 * j + 2 * st] : 0.0, cs = rev ? S.sc[6 * j + 2 * st + 1] : 1.0;
        const double pris = rev ? 0.0 : value_of(qe);
        double Av[3], Bv[3], Cv[3];                     // the axis column of the parent's rotation and its cyclic successors
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          Av[rr] = (a == 0) ? X[rr] : ((a == 1) ? Y[rr] : Z[rr]);
          Bv[rr] = (a == 0) ? Y[rr] : ((a == 1) ? Z[rr] : X[rr]);
          Cv[rr] = (a == 0) ? Z[rr] : ((a == 1) ? X[rr] : Y[rr]);
        }
        double nB[3], nC[3];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          nB[rr] = cs * Bv[rr] + sn * Cv[rr];
          nC[rr] = cs * Cv[rr] - sn * Bv[rr];
          p[rr] = p[rr] + Av[rr] * (t0 + pris) + Bv[rr] * t1 + Cv[rr] * t2;
        }
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          const double na = Av[rr], nb = nB[rr], nc = nC[rr];
          X[rr] = (a == 0) ? na : ((a == 1) ? nc : nb);
          Y[rr] = (a == 0) ? nb : ((a == 1) ? na : nc);
          Z[rr] = (a == 0) ? nc : ((a == 1) ? nb : na);
        }
        double col[6];                                  // this joint's column: revolute (p x axis, axis), prismatic (axis, 0)
        if (rev) {
          double cr[3];
          cross3(p, Av, cr);
          col[0] = cr[0]; col[1] = cr[1]; col[2] = cr[2]; col[3] = Av[0]; col[4] = Av[1]; col[5] = Av[2];
        } else { col[0] = Av[0]; col[1] = Av[1]; col[2] = Av[2]; col[3] = 0.0; col[4] = 0.0; col[5] = 0.0; }
        add_col(col);
      }
    }
    // det of the symmetric positive semi-definite G by elimination without pivoting (det6_spd, on the packed upper triangle)
    double m[6][6];
    {
      int t = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int bb = a; bb < 6; ++bb) { m[a][bb] = G[t]; m[bb][a] = G[t]; ++t; }
    }
    double det = 1.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double piv = m[c][c];
      det *= piv;
      const double ip = (piv > 0.0) ? 1.0 / piv : 0.0;
#pragma unroll
      for (int rr = c + 1; rr < 6; ++rr) {
        const double ff = m[rr][c] * ip;
#pragma unroll
        for (int kx = c + 1; kx < 6; ++kx) m[rr][kx] = fma(-ff, m[c][kx], m[rr][kx]);
      }
    }
    f = sqrt(det > 0.0 ? det : 0.0);
  }
  if (side == 0) f1 = f; else f2 = f;
  }
  // u by DoF: the sweep's central difference, the PREV value where the loop skips the DoF (HYBRID), else 0
  if (grp) { S.uo[lane] = 0.0; if (lane + 21 < 32) S.uo[lane + 21] = 0.0; S.f[lane] = 0.0; S.f[lane + 21] = 0.0; }
  WSYNC();
  if (on) { S.uo[my_i & 31] = 0.5 * (f1 - f2) / dq; S.f[my_i & 31] = 1.0; }      // (f [32]: flags "DoF swept" here)
  WSYNC();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int d = lane + 21 * i;
    if (!grp || d >= NQ) continue;
    if (d < NV) {
      double u = 0.0;
      if (d < nv && ((P.mp_prevmode >> d) & 1u)) u = S.q[d < 6 ? d : d + 1];
      if (d < nv && S.f[d & 31] != 0.0) u = S.uo[d & 31];
      if (A.u && valid) A.u[(size_t)b * NV + d] = (d < nv) ? u : 0.0;
    }
    if (A.q_after && valid) {
      const double a0 = S.q[d & 31];
      A.q_after[(size_t)b * NQ + d] = (d < nq) ? (((P.mp_all >> d) & 1u) ? (a0 + dq) - (dq * 2) : a0) : 0.0;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The tail of runWBC: updateState(joint_config, base_config, running=True) (Robot_Wrapper4.py:1397-1399, 387-428) with
// trunkWorldPos (:1297-1327). One instance per wave. In a rollout the same wave then applies the side effects qpb() has on
// the controller's reference state (:1151-1152, :995-996) and moves the targets one step along their segment.
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) USmem {
  double oMi[24 * 12];
  double q[32];
  double pf[WBC_MAX_FRAMES * 3];
  double ft[16];
};
__global__ void __launch_bounds__(64) wbc_update_kernel(const UpdateArgs A, const DevModel* __restrict__ models,
                                                        const WbcConfig* __restrict__ cfgs) {
  __shared__ USmem U;
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int mid = model_index(A.model_id, b, A.n_models);
  const DevModel& M = models[mid];
  const WbcConfig& cfg = cfgs[mid];
  // ---- every global read of the wave is issued here, before the FK (one memory round trip instead of a chain of them)
  // config = [current base xyz, base_config (IMU quaternion), new joint angles]  (:388-389)
  const int nq = M.nq;
  const bool warm = A.mode == WBC_ROLLOUT_WARMUP;   // updateState(new_config, feedback=False, running=False): the state is q_next as it is
  double c = 0.0;
  if (warm) { if (lane < nq) c = A.q_next[(size_t)b * NQ + lane]; }
  else if (lane < 3) c = A.q_cur[(size_t)b * NQ + lane];
  else if (lane < 7) c = A.imu ? A.imu[(size_t)b * 4 + (lane - 3)] : A.q_next[(size_t)b * NQ + lane];
  else if (lane < nq) c = A.q_next[(size_t)b * NQ + lane];
  const double ft = (lane < 12) ? A.foot_targets[(size_t)b * 15 + lane] : 0.0;
  const double eet = (A.ee_target && lane < 15) ? A.ee_target[(size_t)b * 15 + lane] : 0.0;
  const double ees = (A.ee_target && A.ee_step && lane < 15) ? A.ee_step[(size_t)b * 15 + lane] : 0.0;
  const double rref = (A.ee_prev_rot && A.ee_ref_rot && lane < 45) ? A.ee_ref_rot[(size_t)b * 45 + lane] : 0.0;
  const double tt = (A.trunk_target && lane < 3) ? A.trunk_target[(size_t)b * 3 + lane] : 0.0;
  const double tts = (A.trunk_target && A.trunk_step && lane < 3) ? A.trunk_step[(size_t)b * 3 + lane] : 0.0;
  const double ter = (A.trunk_prev_rot && A.trunk_ref_euler && lane < 3) ? A.trunk_ref_euler[(size_t)b * 3 + lane] : 0.0;
  int st = 0, stm = 0, it = 0, its = 0;
  if (lane == 0) {
    if (A.status_max) { st = A.status[b]; stm = A.status_max[b]; }
    if (A.iters_sum) { it = A.iters[b]; its = A.iters_sum[b]; }
  }
  const LaneConst lc = load_lane_const(M, cfg, lane);
  if (lane < 32) U.q[lane] = c;
  if (lane < 12) U.ft[lane] = ft;
  WSYNC();
  fk_levels(U.oMi, U.q, load_hdr(M), lc, lane);
  if (lane < M.nframes) {
    const double* Pj = U.oMi + lc.fj_off;
#pragma unroll
    for (int r = 0; r < 3; ++r) U.pf[3 * lane + r] = Pj[9 + r] + Pj[r] * lc.f0 + Pj[3 + r] * lc.f1 + Pj[6 + r] * lc.f2;
  }
  WSYNC();
  // trunkWorldPos: trunk_pos = WPA - WRB . BPA  (:1321-1325), evaluated uniformly
  const double* Pt = U.oMi + 12 * M.frame_joint[WBC_FR_TRUNK];   // R column-major
  double WPA[3], BPA[3], base[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double t = U.pf[3 * WBC_FR_TRUNK + i];
    WPA[i] = (U.ft[i] + U.ft[3 + i] + U.ft[6 + i] + U.ft[9 + i]) / 4;
    BPA[i] = ((U.pf[i] - t) + (U.pf[3 + i] - t) + (U.pf[6 + i] - t) + (U.pf[9 + i] - t)) / 4;
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) base[r] = WPA[r] - (Pt[r] * BPA[0] + Pt[3 + r] * BPA[1] + Pt[6 + r] * BPA[2]);
  if (warm) { base[0] = U.q[0]; base[1] = U.q[1]; base[2] = U.q[2]; }   // no estimator while warming up (running == False, :414)
  if (lane < NQ) A.q_new[(size_t)b * NQ + lane] = (lane == 0) ? base[0] : (lane == 1) ? base[1] : (lane == 2) ? base[2] : c;
  if (A.grip_trace && lane < 3) {
    // gripper_bar after the base correction: the whole tree translates rigidly with the base
    const double d = (lane == 0) ? base[0] - U.q[0] : (lane == 1) ? base[1] - U.q[1] : base[2] - U.q[2];
    A.grip_trace[(size_t)b * 3 + lane] = U.pf[3 * (WBC_FR_EE0 + 4) + lane] + d;
  }
  if (lane == 0) {
    if (A.status_max && st > stm) A.status_max[b] = st;
    if (A.iters_sum) A.iters_sum[b] = its + it;
  }
  // ---- side effects of qpb() on the reference state, then the targets move on
  if (A.ee_target && lane < 15) {
    const int e = lane / 3;
    const size_t i = (size_t)b * 15 + lane;
    if (cfg.task_ee[e] && A.prev_ee_target) A.prev_ee_target[i] = eet;               // prev_EE_pos[i] = target (:1151)
    if (A.ee_step) A.ee_target[i] = eet + ees;
  }
  if (A.ee_prev_rot && A.ee_ref_rot && lane < 45) {
    const int e = lane / 9;
    if (cfg.task_ee[e]) A.ee_prev_rot[(size_t)b * 45 + lane] = rref;                 // prev_EE_CoM_rot[i] = R* (:1152)
  }
  if (A.trunk_target && lane < 3) {
    const size_t i = (size_t)b * 3 + lane;
    if (cfg.task_trunk && A.prev_trunk_target) A.prev_trunk_target[i] = tt;          // prev_trunk_ref = target (:995)
    if (A.trunk_step) A.trunk_target[i] = tt + tts;
  }
  if (cfg.task_trunk && A.trunk_prev_rot && A.trunk_ref_euler) {                      // old_ref_trunk_rot_matrix = R* (:996)
    const SinCos a = sincos_cw(rdl(ter, 0)), bb = sincos_cw(rdl(ter, 1)), cc = sincos_cw(rdl(ter, 2));
    double Rs[9];
    Rs[0] = cc.c * bb.c; Rs[1] = cc.c * bb.s * a.s - cc.s * a.c; Rs[2] = cc.c * bb.s * a.c + cc.s * a.s;
    Rs[3] = cc.s * bb.c; Rs[4] = cc.s * bb.s * a.s + cc.c * a.c; Rs[5] = cc.s * bb.s * a.c - cc.c * a.s;
    Rs[6] = -bb.s;       Rs[7] = bb.c * a.s;                     Rs[8] = bb.c * a.c;
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 9; ++i) A.trunk_prev_rot[(size_t)b * 9 + i] = Rs[i];
    }
  }
}

// One instance on the general path, called by the packed kernels for what they cannot reduce themselves (the TAIL: a stance-leg block of rank < 2,
// a flagged block on the orth kernel). It reads the kernel's argument block AGAIN, through the kernarg segment pointer (KernelArgs is the first
// kernel parameter of both callers): handed the caller's own `A`, the general path's ~90 scalars were fetched at kernel entry and kept alive —
// spilled to VGPR lanes — across the whole packed path: 480 extra v_writelane / v_readlane in the common path, 3 % of the step (same-box A/B,
// tools/ab_bench.sh). A real call is not an option: arguments arrive in VGPRs, and the general path pins configuration scalars to SGPRs.
template <bool WARM, bool ORTH>
__device__ __forceinline__ void tail_instance(Smem* Sp, const int bt_v, const DevModel* __restrict__ models, const WbcConfig* __restrict__ cfgs,
                                              const DevPlan* __restrict__ plans) {   // (the kernel's own noalias table pointers: scalar loads)
  Smem& S = *Sp;
  __attribute__((address_space(4))) const KernelArgs* Ap =
      (const __attribute__((address_space(4))) KernelArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(Ap));                                   // (opaque: these loads are not merged with, nor hoisted to, the kernel's entry loads)
  const __attribute__((address_space(4))) KernelArgs& A = *Ap;   // (kept in the constant address space: scalar loads)
  const int bt_ = __builtin_amdgcn_readfirstlane(bt_v);
  const bool has2 = A.in.trunk_target || A.in.prev_trunk_target || A.in.trunk_ref_euler || A.in.trunk_prev_rot ||
                    A.in.com_target || A.in.com_target_vel;
  const bool has3 = A.in.ee_ref_rot != nullptr;
  int ln = threadIdx.x;
  asm volatile("" : "+v"(ln));
  WSYNC();
  S.cl[ln] = 0.0;
  const int mi = model_index(A.in.model_id, bt_, A.n_models);
  const InRegs cur = load_inputs(A.in, bt_, ln, has2, has3);
  const LaneConst lc = load_lane_const(models[mi], cfgs[mi], ln);
  stage_inputs(S, cur, ln, has2, has3);
  WSYNC();
  process_instance<MODE_TICK, WARM, ORTH>(S, A, models[mi], cfgs[mi], plans[mi], lc, cur, bt_, ln, 0ull);
  WSYNC();
}

// ================================================================================================
// The PACKED sim3-tick kernel: FOUR robot instances per wavefront, one per 16-lane DPP row.
//
// The compact kernel above keeps one instance per wave, and its reduced QP (n' = 11 unknowns, <= 16 rows) lights 11-16 of the 64
// lanes: 3.3 k VALU wave-instructions per tick for ~1.4e4 useful flops. Here lane = 16 r + s: instance r of the wave, s = reduced
// variable / constraint row / FK slot. Every stage is written for 16 lanes:
//   FK          level-synchronous over a per-plan schedule (DevPlan.pk_fk: at most five joints per tree level — four legs + the
//               arm chain), sin/cos of the joint angles computed beforehand two per lane;
//   columns     lane s owns the WORLD Jacobian column of reduced variable s (task rows, trunk-box rows) and of eliminated leg DoF
//               s < 12 (contact rows -> K_e, velocity bounds);
//   assembly    row s of H' accumulated straight into registers from the task image At (LDS), G = -K^-1 B on lanes s < 12;
//   QP          the dual active-set of qp_core with per-row state: reductions are DPP row butterflies (no v_readlane), a value
//               at a row-dependent lane comes through ds_bpermute, the Cholesky column is broadcast through a per-instance LDS
//               vector, control flow is per-row predication with the loops running to the slowest of the four instances.
// Applies to the sim3 switch-set family only (launch_tick_auto): Grip task or none, optionally the trunk task (TRUNK variant), posture PREV /
// Tikhonov / static HYBRID, trunk box + foot contacts, velocity bounds on, no CoM rows; working sets in and out on the WARM variant; the gripper's
// orientation reference is honoured. A rank-deficient leg block is pivoted in place (the swap); instances with a
// leg block of rank < 2 are redone on the general path by their own wave at the end of this kernel (the tail: tail_instance). Same arithmetic per
// instance as process_sim3.
// ================================================================================================
constexpr int PLD = 14;                     // row stride of the matrices (even: rows are 16-byte aligned for ds_read_b128; 7 s mod 16 is a
                                            // permutation, so "lane = row" b128 reads of two instances interleave conflict-free)
constexpr int PN = 16;                      // lanes = constraint rows per instance
constexpr int PV = 12;                      // reduced variables per instance the packed kernel is compiled for (n' = 11 / 10 here)
struct __attribute__((aligned(16))) PInst {
  double M1[PV * PLD];                      // oMi scratch (runs on into M2: 22 joints x 12 doubles) -> T = R^-1
  double M2[PV * PLD];                      // ... sin / cos table in its tail during FK; then At [16][6], K / B scratch -> J
  double Cq[PN * 6];                        // reduced constraint rows x base columns (all the reduced rows touch the base only);
                                            // rows p_keep + l are the rows of G (eliminated leg DoF l x base DoF)
  double pad_[8];
};
struct __attribute__((aligned(16))) PVec {
  double in[40];                            // q [27], gripper target [3] @28, previous [3] @31, trunk box centre [4] @34
  double xv[PN], dv[PN], yv[PN], tv[PN];
  double cl[32];                            // row-bound staging -> Cholesky column broadcast (entries 12..31 zero) -> qdot by DoF
  double pad_[8];
};
// Bank placement (ds_read_b64 / b128 bank = dword address mod 64; the four instances of a wave issue every access together): the vectors are read
// as broadcasts or "lane = element" b64, which collide when the instances sit a multiple of the 256-byte bank row apart and are conflict-free
// 128 B (mod 256) apart. The matrix blocks sat a multiple of 256 B apart in round 2 (measured best for the "lane = row" b128 reads then);
// with the broadcast row reads the kernel has since (At / Cq rows in the H' accumulation, the violation scan and normal_d: all lanes of an
// instance on one address, four instances on four) that distance made all four meet in one bank group — 192 B (mod 256) apart measures
// +1.3 % on the benchmark (same-box A/B, four rounds: 330.1 vs 325.9 M ticks/s; 160 B apart +0.6 %).
static_assert(sizeof(PInst) % 256 == 192, "matrix blocks: 192 B (mod the 256-byte bank row) apart");
static_assert(sizeof(PVec) % 256 == 128, "vector blocks: half a bank row apart (mod 256 B)");
struct __attribute__((aligned(16))) SmemP { PInst I[4]; PVec V[4]; };

__device__ __forceinline__ double rsum16(double v) {     // sum over the lane's 16-lane row, result in every lane of the row
  v += dpp<DPP_XOR1>(v); v += dpp<DPP_XOR2>(v); v += dpp<DPP_HALF_MIRROR>(v); v += dpp<DPP_MIRROR>(v);
  return v;
}
__device__ __forceinline__ double rmin16(double v) {
  v = fmin(v, dpp<DPP_XOR1>(v)); v = fmin(v, dpp<DPP_XOR2>(v)); v = fmin(v, dpp<DPP_HALF_MIRROR>(v)); v = fmin(v, dpp<DPP_MIRROR>(v));
  return v;
}
__device__ __forceinline__ double bperm(double v, int src_lane) {     // v of lane src_lane (any lane index 0..63, per lane)
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int bpermi(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_or(unsigned long long v) {
  int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
  lo |= __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  hi |= __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ unsigned long long ror16(unsigned long long v) {   // bitwise OR over the lane's 16-lane row
  v = dpp_or<DPP_XOR1>(v); v = dpp_or<DPP_XOR2>(v); v = dpp_or<DPP_HALF_MIRROR>(v); v = dpp_or<DPP_MIRROR>(v);
  return v;
}

#ifdef WBC_ABLATE
#define PSTOP(k, val) do { if (A.dbg_stop == 100 + (k)) { if (valid) { A.out.qdot[(size_t)b * NV + s] = (val); if (s == 0) A.out.status[b] = 0; } return; } } while (0)
#else
#define PSTOP(k, val) do { } while (0)
#endif
// WARM: the variant that takes / returns working sets (warm start, KernelArgs.ws_in / ws_out: the analogue of qpOASES' hotstart,
// QP_Wrapper.py:55-73); the cold variant carries no trace of it. The seeds go through the dual method's own ADD step (Householder on
// J2, column (-T r / delta, 1 / delta) of T) without its search / ratio test / partial steps, the iterate and the multipliers are then
// rebuilt from the factors (x = x0 + J1 w, u = T w, w = T's with s the seeds' slacks at the unconstrained minimiser x0), seeds with
// a negative multiplier are dropped again (restoration), and the dual iterations carry on from that S-pair: qp_core<.., WARM>'s
// scheme (tests/gi_variant.py solve_v3) for a problem without equalities.
// TRUNK: the variant that carries the trunk task (trunkA / calcTargetVelTrunk2, Robot_Wrapper4.py:487-490, 948-1015): six more task rows on
// the base columns; its inputs and parameters are staged in vectors that are free until the contact stage, so the common variant's
// register allocation is untouched.
// QCON: the variant for a caller's (or wbc_posture_par_kernel's) posture target `posture_u` and constraint state `q_con` — qpJointb "MANI" / literal
// "HYBRID" with sweeps that matter (Robot_Wrapper4.py:1220-1260): the tasks are formed at q, then the kinematics are redone at
// q_con and the contact rows, the trunk box, the damper bounds and the integration see THAT state (a second FK pass, as in process_instance).
template <bool WARM, bool TRUNK = false, bool QCON = false>
__global__ void __launch_bounds__(64, 2) wbc_tick_sim3p_kernel(const KernelArgs A, const DevModel* __restrict__ models,
                                                               const WbcConfig* __restrict__ cfgs, const DevPlan* __restrict__ plans) {
  // (the general kernel's layout shares the allocation: an instance this kernel cannot reduce — a stance-leg block of rank < 2 — is
  //  redone on the general path by the SAME wave at the end, see the tail; both layouts leave 8 waves per CU)
  __shared__ union { SmemP P; Smem G; } SU;
  static_assert(sizeof(Smem) <= 20480 && sizeof(SmemP) <= 20480, "8 waves per CU");
  SmemP& SP = SU.P;
  const int lane = threadIdx.x, r = lane >> 4, s = lane & 15, rbase = lane & 48;
  PInst& I = SP.I[r];
  PVec& V = SP.V[r];
  const int b_raw = 4 * blockIdx.x + r;
  const bool valid = b_raw < A.B;
  const int b = valid ? b_raw : A.B - 1;
  int mid = 0;
  if (A.in.model_id) { mid = A.in.model_id[b]; mid = mid < 0 ? 0 : (mid >= A.n_models ? A.n_models - 1 : mid); }
  const DevModel& M = models[mid];
  const WbcConfig& cfg = cfgs[mid];
  const DevPlan& P = plans[mid];
  const double dt = A.dt, inv_dt = 1.0 / A.dt;

  // ---- loads: inputs (coalesced per instance), then the per-lane tables
  {
    const double* qg = A.in.q + (size_t)b * NQ;
    const double q0 = qg[s], q1 = (16 + s < NQ) ? qg[16 + s] : 0.0;
    double ex = 0.0;
    if (s < 3) ex = A.in.ee_target ? A.in.ee_target[(size_t)b * 15 + 12 + s] : 0.0;
    else if (s < 6) ex = A.in.prev_ee_target ? A.in.prev_ee_target[(size_t)b * 15 + 12 + (s - 3)] : 0.0;
    else if (s < 10) ex = A.in.trunk_box_center ? A.in.trunk_box_center[(size_t)b * 4 + (s - 6)] : 0.0;
    V.in[s] = q0;
    if (16 + s < 28) V.in[16 + s] = q1;
    if (s < 10) V.in[28 + s] = ex;
    V.cl[s] = 0.0; V.cl[16 + s] = 0.0;
    if (TRUNK) {   // trunk_target [3], prev_trunk_target [3], trunk_ref_euler [3], trunk_prev_rot [9] -> V.tv [16] + V.xv [0..1]; the configuration's
                   // trunk_W [6], trunk_w, trunk_gain [6] -> V.xv [2..14] (both vectors are free until the contact stage)
      auto tin = [&](const int k) -> double {
        return (k < 3) ? A.in.trunk_target[(size_t)b * 3 + k] : (k < 6) ? A.in.prev_trunk_target[(size_t)b * 3 + (k - 3)]
             : (k < 9) ? A.in.trunk_ref_euler[(size_t)b * 3 + (k - 6)] : A.in.trunk_prev_rot[(size_t)b * 9 + (k - 9)];
      };
      const double t0 = tin(s), t1 = (s < 2) ? tin(16 + s) : 0.0;
      const double tw = (s < 13) ? (&cfg.trunk_W[0])[s] : 0.0;
      V.tv[s] = t0;
      if (s < 2) V.xv[s] = t1;
      if (s < 13) V.xv[2 + s] = tw;
    }
    if (WARM && s < 2) {                  // the carried working set: two words per instance, parked (as bit patterns) in V.in[38..39]
      const unsigned long long w = (A.ws_in && valid) ? A.ws_in[2 * (size_t)b + s] : 0ull;
      V.in[38 + s] = __longlong_as_double((long long)w);
    }
    if (A.in.ee_ref_rot) {                // the gripper's orientation reference and its previous value (free vectors until the QP)
      if (s < 9) { V.dv[s] = A.in.ee_ref_rot[(size_t)b * 45 + 36 + s]; V.yv[s] = A.in.ee_prev_rot[(size_t)b * 45 + 36 + s]; }
    }
  }
  if (TRUNK) {
    WSYNC();                               // (the staged inputs are visible)
    const double* const qv = V.in;
    // calcTargetVelTrunk2 (Robot_Wrapper4.py:948-1015) / TrunkB (:914-920): the trunk frame is the free-flyer's own placement (the plan checks
    // it), so the target velocity depends on the inputs alone — formed here, where hardly anything is live
    const double* tw = V.xv + 2;           // trunk_W [0..5], trunk_w [6], trunk_gain [7..12]
    const double* xt = V.tv;
    const double* xp = V.tv + 3;
    const double* er = V.tv + 6;
    double* const sh = I.M2;               // (free until the FK)
    double Rt_[9], fq[4], rq[4], Rs[9], vel[6];
    quat_to_R(qv + 3, Rt_);
    R_to_quat(Rt_, fq);
    {
      const SinCos t = sincos_cw(s < 3 ? er[s < 3 ? s : 0] : 0.5 * er[(s < 6 ? s : 3) - 3]);   // reference angles and their halves, one per lane
      if (s < 6) { sh[2 * s] = t.s; sh[2 * s + 1] = t.c; }
      WSYNC();
      const double sa = sh[0], ca = sh[1], sb = sh[2], cb = sh[3], sc_ = sh[4], cc = sh[5];
      Rs[0] = cc * cb; Rs[1] = cc * sb * sa - sc_ * ca; Rs[2] = cc * sb * ca + sc_ * sa;
      Rs[3] = sc_ * cb; Rs[4] = sc_ * sb * sa + cc * ca; Rs[5] = sc_ * sb * ca - cc * sa;
      Rs[6] = -sb;      Rs[7] = cb * sa;                 Rs[8] = cb * ca;
      const double qx[4] = {sh[6], 0, 0, sh[7]}, qy[4] = {0, sh[8], 0, sh[9]}, qz[4] = {0, 0, sh[10], sh[11]};
      double tq[4];
      quat_mul(qy, qx, tq);
      quat_mul(qz, tq, rq);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) vel[i] = (xt[i] - xp[i]) * inv_dt + tw[7 + i] * ((xt[i] - qv[i]) * inv_dt);
    const double qe0 = fq[3] * rq[0] - fq[0] * rq[3] + fq[1] * rq[2] - fq[2] * rq[1];   // :974
    const double qe1 = fq[3] * rq[1] - fq[1] * rq[3] - fq[0] * rq[2] + fq[2] * rq[0];   // :975
    const double qe2 = fq[3] * rq[2] - fq[3] * rq[2] + fq[0] * rq[1] - fq[1] * rq[0];   // :976 (sic)
    const double Ro[9] = {V.tv[9], V.tv[10], V.tv[11], V.tv[12], V.tv[13], V.tv[14], V.tv[15], V.xv[0], V.xv[1]};
    double D[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) D[i] = (Rs[i] - Ro[i]) * inv_dt;
    // skew = D Rs (R*, not R*^T: :984); omega = (S[2][1], S[0][2], S[1][0]) + K qe
    vel[3] = (D[6] * Rs[1] + D[7] * Rs[4] + D[8] * Rs[7]) + tw[10] * qe0;
    vel[4] = (D[0] * Rs[2] + D[1] * Rs[5] + D[2] * Rs[8]) + tw[11] * qe1;
    vel[5] = (D[3] * Rs[0] + D[4] * Rs[3] + D[5] * Rs[6]) + tw[12] * qe2;
    const double trunk_w = tw[6];
    // trunkA (Robot_Wrapper4.py:487-490, WORLD): the task's rows live on the six base columns, and the free-flyer's own Jacobian columns are its
    // placement (linear DoF c: column c of R; angular DoF c: p x column c, column c) — so the task's WHOLE contribution, the 6 x 6 block of H'
    // and its part of g, is formed here, where hardly anything is live, and parked in Cq [0..41] (free until the constraint stage). In the task
    // stage it used to keep 30 values alive across the Grip block: ~55 spill instructions in the hot path, 0.14 ms per 65536 ticks.
    double at[6] = {0, 0, 0, 0, 0, 0};
    {
      const int c = s < 3 ? s : (s < 6 ? s - 3 : 0);
      const double col[3] = {Rt_[c], Rt_[3 + c], Rt_[6 + c]};
      const double pr[3] = {qv[0], qv[1], qv[2]};
      double lin[3] = {col[0], col[1], col[2]}, ang[3] = {0, 0, 0};
      if (s >= 3) { ang[0] = col[0]; ang[1] = col[1]; ang[2] = col[2]; cross3(pr, ang, lin); }
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        at[rr] = (s < 6) ? (tw[rr] * lin[rr]) * trunk_w : 0.0;
        at[3 + rr] = (s < 6) ? (tw[3 + rr] * ang[rr]) * trunk_w : 0.0;
      }
    }
    double* const At2 = I.M1;              // [6][6] (free until the FK)
    WSYNC();                               // (everyone has read the gains)
    if (s < 6) {
#pragma unroll
      for (int rr = 0; rr < 6; rr += 2) sts2(At2 + s * 6 + rr, at[rr], at[rr + 1]);
    }
    WSYNC();
    if (s < 6) {
      double gs = 0.0;
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) gs = fma(-at[rr], vel[rr] * trunk_w, gs);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const double2a t0 = lds2(At2 + k * 6), t1 = lds2(At2 + k * 6 + 2), t2 = lds2(At2 + k * 6 + 4);
        I.Cq[s * 6 + k] = fma(at[0], t0.x, fma(at[1], t0.y, fma(at[2], t1.x, fma(at[3], t1.y, fma(at[4], t2.x, at[5] * t2.y)))));
      }
      I.Cq[36 + s] = gs;
    }
    WSYNC();
    __builtin_amdgcn_sched_barrier(0);    // (the per-lane records below are fetched after this block: live across it they spilled 110 VGPRs)
  }
  const int nv = M.nv, nq = M.nq, n = P.n_red, nelim = P.nelim, nl = 3 * nelim, p_keep = P.p_keep, p = p_keep + nl;
  const unsigned fl = P.flags;
  const bool c_con_trunk = fl & 2u;
  const int c_task_joint = (fl >> 4) & 7u;
  const bool has_grip = (P.task_ee_mask >> 4) & 1u;
  // per-lane records (one load level: nothing waits for an index): reduced variable s, eliminated leg DoF s, first FK level
  const DevPlan::PkCol cv = P.pk_var[s], cg = P.pk_leg[s];
  DevPlan::PkJoint fkn = P.pk_fk[0][s];
  const int scq0 = P.pk_scq[(2 + s) & 31], scq1 = P.pk_scq[(18 + s) & 31];
  const int dof0 = cv.dof, dof1 = cg.dof;
  const int c0_joint = cv.joint, c0_lin = cv.lin, c0_ang = cv.ang, c1_joint = cg.joint, c1_lin = cg.lin, c1_ang = cg.ang;
  const int dq0 = cv.dq_idx, dq1 = cg.dq_idx;
  const double dlo0 = cv.d_lo, dhi0 = cv.d_hi, dvm0 = cv.d_vm, dlo1 = cg.d_lo, dhi1 = cg.d_hi, dvm1 = cg.d_vm;
  const double dcoef = cfg.damper_coef, dqi = cfg.damper_qi, dqs = cfg.damper_qs;
  const int gj = M.frame_joint[WBC_FR_EE0 + 4];
  const double gp0 = M.frame_p[WBC_FR_EE0 + 4][0], gp1 = M.frame_p[WBC_FR_EE0 + 4][1], gp2 = M.frame_p[WBC_FR_EE0 + 4][2];
  const unsigned gsup = P.redsup[WBC_FR_EE0 + 4];
  const double ee_w = cfg.ee_w[4];
  double eW[6], eG[3];
#pragma unroll
  for (int i = 0; i < 6; ++i) eW[i] = cfg.ee_W[4][i];
#pragma unroll
  for (int i = 0; i < 3; ++i) eG[i] = cfg.ee_gain[4][i];
  const double joint_w = cfg.joint_w, tb_z = cfg.trunk_box_z_frac, tb_a = cfg.trunk_box_ang, tb_s = cfg.trunk_box_scale;
  WSYNC();
  const double* const qv = V.in;
  PSTOP(6, qv[s] + dlo0 + dlo1 + eW[0] + (double)(fkn.joint + scq0 + scq1));
  // ---- sin / cos of the joint angles: joint j (>= 2) reads q[idx_q[j]]; two joints per lane
  double* const oMi = I.M1;                 // [22][12], runs on into M2
  double* const sc = I.M2 + PV * PLD - 48;  // sin / cos table: the tail of M2, free until J is written
  {
    if (scq0 >= 0) { const SinCos t = sincos_cw(qv[scq0]); sc[2 * (2 + s)] = t.s; sc[2 * (2 + s) + 1] = t.c; }
    if (scq1 >= 0) { const SinCos t = sincos_cw(qv[scq1]); sc[2 * (18 + s)] = t.s; sc[2 * (18 + s) + 1] = t.c; }
    // root free-flyer (joint 1): R from the quaternion exactly as Eigen's toRotationMatrix, p = xyz; R column-major then p
    if (s == 0) {
      double Rt[9];
      quat_to_R(qv + 3, Rt);
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) oMi[12 + 3 * c + rr] = Rt[3 * rr + c];
      oMi[12 + 9] = qv[0]; oMi[12 + 10] = qv[1]; oMi[12 + 11] = qv[2];
    }
  }
  WSYNC();
  PSTOP(7, oMi[12 + s] + sc[4 + s]);
  // ---- P1: pin.forwardKinematics, level by level (Robot_Wrapper4.py:400). The level's joint and its constants are fetched
  // inside the loop (L1-resident tables): kept live for all five levels they cost 60 VGPRs
#pragma unroll 1
  for (int L = 0; L < 5; ++L) {
    const DevPlan::PkJoint fk = fkn;
    if (L + 1 < 5) fkn = P.pk_fk[L + 1][s];          // next level's record is on its way while this level is computed
    const int j = fk.joint;
    if (j >= 0) {
      const bool rev = fk.rev != 0;
      const int a0 = fk.a0, a1 = fk.a1, a2 = fk.a2;
      const double* Pp = oMi + 12 * fk.parent;
      const double sn = rev ? sc[2 * j] : 0.0, cs = rev ? sc[2 * j + 1] : 1.0;
      const double pris = rev ? 0.0 : qv[fk.q_idx];
      double Av[3], Bv[3], Cv[3], Pv[3];
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) { Av[rr] = Pp[a0 + rr]; Bv[rr] = Pp[a1 + rr]; Cv[rr] = Pp[a2 + rr]; Pv[rr] = Pp[9 + rr]; }
      double* Po = oMi + 12 * j;
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        Po[a0 + rr] = Av[rr];
        Po[a1 + rr] = cs * Bv[rr] + sn * Cv[rr];
        Po[a2 + rr] = cs * Cv[rr] - sn * Bv[rr];
        Po[9 + rr] = Pv[rr] + Av[rr] * (fk.t0 + pris) + Bv[rr] * fk.t1 + Cv[rr] * fk.t2;
      }
    }
    WSYNC();
  }
  // ---- P3: Jacobian columns (WORLD): of reduced variable s, and (linear part) of eliminated leg DoF s
  PSTOP(1, oMi[12 * 4 + (s & 7)]);
  double lin0[3] = {0, 0, 0}, ang0[3] = {0, 0, 0}, lin1[3] = {0, 0, 0};
  if (s < n) {
    const double* Pj = oMi + 12 * c0_joint;
    const double pj[3] = {Pj[9], Pj[10], Pj[11]};
    if (c0_ang >= 0) { ang0[0] = Pj[3 * c0_ang]; ang0[1] = Pj[3 * c0_ang + 1]; ang0[2] = Pj[3 * c0_ang + 2]; cross3(pj, ang0, lin0); }
    if (c0_lin >= 0) { lin0[0] = Pj[3 * c0_lin]; lin0[1] = Pj[3 * c0_lin + 1]; lin0[2] = Pj[3 * c0_lin + 2]; }
  }
  if (s < nl) {
    const double* Pj = oMi + 12 * c1_joint;
    const double pj[3] = {Pj[9], Pj[10], Pj[11]};
    if (c1_ang >= 0) { const double a1[3] = {Pj[3 * c1_ang], Pj[3 * c1_ang + 1], Pj[3 * c1_ang + 2]}; cross3(pj, a1, lin1); }
    if (c1_lin >= 0) { lin1[0] = Pj[3 * c1_lin]; lin1[1] = Pj[3 * c1_lin + 1]; lin1[2] = Pj[3 * c1_lin + 2]; }
  }
  // trunk frame = the root joint's placement (imu frame: identity offset); gripper_bar origin
  double Rtr[9], ptr[3], pfe[3];
  {
    const double* Pr = oMi + 12 * M.frame_joint[WBC_FR_TRUNK];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) Rtr[3 * rr + c] = Pr[3 * c + rr];
    ptr[0] = Pr[9]; ptr[1] = Pr[10]; ptr[2] = Pr[11];
    const double* Pg = oMi + 12 * gj;
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) pfe[rr] = Pg[9 + rr] + Pg[rr] * gp0 + Pg[3 + rr] * gp1 + Pg[6 + rr] * gp2;
  }
  WSYNC();   // oMi is dead: M1 / M2 are free

  // ---- task stack (qpA / qpb, Robot_Wrapper4.py:1271-1294): Grip rows of reduced variable s -> At[s][6]; g
  double* const At = I.M2;                 // [16][6]
  double* const Kb = I.M2 + 16 * 6;        // [12][4]: linear WORLD column of leg DoF l
  double* const Bb = I.M2 + 16 * 6 + 48;   // [6][4]:  linear WORLD column of base DoF c
  double g = 0.0;
  double a[6] = {0, 0, 0, 0, 0, 0};
  if (has_grip) {
    const bool sup = (s < n) && ((gsup >> s) & 1u);
    double wxp[3];
    cross3(ang0, pfe, wxp);                // endEffectorA2 (:474-484): LOCAL_WORLD_ALIGNED = lin + ang x p_f
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
      a[rr] = sup ? eW[rr] * ((lin0[rr] + wxp[rr]) * ee_w) : 0.0;
      a[3 + rr] = sup ? eW[3 + rr] * (ang0[rr] * ee_w) : 0.0;
    }
    const double* xt = V.in + 28;
    const double* xp = V.in + 31;
#pragma unroll
    for (int i = 0; i < 3; ++i) {          // calcTargetVelEE3 (:1052-1157); EndEffectorB2 (:907-910)
      const double br = ((xt[i] - xp[i]) * inv_dt + eG[i] * ((xt[i] - pfe[i]) * inv_dt)) * ee_w;
      g = fma(-a[i], br, g);
    }
    if (A.in.ee_ref_rot) {                 // omega = vee(((R* - R*_prev)/dt) R*^T)  (:1125-1128, 1133); zero when the reference rests
      const double* Rs = V.dv;
      const double* Rp = V.yv;
      double D[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) D[i] = (Rs[i] - Rp[i]) * inv_dt;
      const double w3 = D[6] * Rs[3] + D[7] * Rs[4] + D[8] * Rs[5];   // S[2][1]
      const double w4 = D[0] * Rs[6] + D[1] * Rs[7] + D[2] * Rs[8];   // S[0][2]
      const double w5 = D[3] * Rs[0] + D[4] * Rs[1] + D[5] * Rs[2];   // S[1][0]
      g = fma(-a[3], w3 * ee_w, g); g = fma(-a[4], w4 * ee_w, g); g = fma(-a[5], w5 * ee_w, g);
    }
  }
  if (TRUNK && s < 6) g += I.Cq[36 + s];    // the trunk task's part (formed at the top)
#pragma unroll
  for (int rr = 0; rr < 6; rr += 2) sts2(At + s * 6 + rr, a[rr], a[rr + 1]);
  if (!QCON) {   // (QCON: the contact rows belong to the constraint state, see the second pass below)
    if (s < nl) { Kb[4 * s] = lin1[0]; Kb[4 * s + 1] = lin1[1]; Kb[4 * s + 2] = lin1[2]; }
    if (s < 6) { Bb[4 * s] = lin0[0]; Bb[4 * s + 1] = lin0[1]; Bb[4 * s + 2] = lin0[2]; }
  }
  // posture rows (qpJointA / qpJointb, :1199-1268) of reduced variable s and of leg DoF s
  const double dpost = (1.0 / nv) * joint_w;
  double g1 = 0.0;                          // posture term of leg DoF s in g
  {
    const bool prev0 = (c_task_joint == WBC_JOINT_PREV) || (c_task_joint >= WBC_JOINT_MANI && !((P.post_zero >> dof0) & 1u));
    const bool prev1 = (c_task_joint == WBC_JOINT_PREV) || (c_task_joint >= WBC_JOINT_MANI && !((P.post_zero >> dof1) & 1u));
    double u0 = (prev0 && s < n) ? qv[dof0 < 6 ? dof0 : dof0 + 1] : 0.0;
    double u1 = (prev1 && s < nl) ? qv[dof1 < 6 ? dof1 : dof1 + 1] : 0.0;
    if (QCON && A.in.posture_u) {           // the posture kernel's (or the caller's) target, by DoF
      u0 = (s < n) ? A.in.posture_u[(size_t)b * NV + dof0] : 0.0;
      u1 = (s < nl) ? A.in.posture_u[(size_t)b * NV + dof1] : 0.0;
    }
    if (s < n) g = fma(-dpost, (1.0 / nv) * u0 * joint_w, g);
    if (s < nl) g1 = -dpost * ((1.0 / nv) * u1 * joint_w);
  }
  if (s >= n) g = 0.0;
  if (A.post_static && P.post_pert) {     // the state qpJointb leaves behind: bounds and integrate see it
    WSYNC();
    if (((P.post_pert >> s) & 1u)) V.in[s] = (qv[s] + 0.0002) - (0.0002 * 2);
    if (16 + s < NQ && ((P.post_pert >> (16 + s)) & 1u)) V.in[16 + s] = (qv[16 + s] + 0.0002) - (0.0002 * 2);
  }
  WSYNC();
  // row s of H' = sum_r At[s][r] At[k][r] (+ posture): straight into the registers the Cholesky sweep works on
  double h[PV];
  {
#pragma unroll
    for (int k = 0; k < PV; ++k) {
      const double2a t0 = lds2(At + k * 6), t1 = lds2(At + k * 6 + 2), t2 = lds2(At + k * 6 + 4);
      h[k] = fma(a[0], t0.x, fma(a[1], t0.y, fma(a[2], t1.x, fma(a[3], t1.y, fma(a[4], t2.x, a[5] * t2.y)))));
    }
#pragma unroll
    for (int k = 0; k < PV; ++k) if (k == s) h[k] += (s < n) ? dpost * dpost : 1.0;   // (lanes >= PV carry an all-zero row: harmless)
    if (TRUNK && s < 6) {   // the trunk task's 6 x 6 block on the base columns (formed at the top)
#pragma unroll
      for (int k = 0; k < 6; k += 2) { const double2a v = lds2(I.Cq + s * 6 + k); h[k] += v.x; h[k + 1] += v.y; }
    }
  }
  PSTOP(2, h[0] + h[5] + h[11] + g);
  if (QCON && A.in.q_con) {
    // ---- the second pass: findConstraints, velDamperJointConstraints and the integration see q_con (the state qpJointb leaves behind); the task
    // image has been consumed (h, g), so M1 / M2 are free for the kinematics again. Same code as the first pass.
    WSYNC();
    {
      const double* qg = A.in.q_con + (size_t)b * NQ;
      const double c0 = qg[s], c1 = (16 + s < NQ) ? qg[16 + s] : 0.0;
      V.in[s] = c0;
      if (16 + s < 28) V.in[16 + s] = c1;
    }
    fkn = P.pk_fk[0][s];
    WSYNC();
  {
    if (scq0 >= 0) { const SinCos t = sincos_cw(qv[scq0]); sc[2 * (2 + s)] = t.s; sc[2 * (2 + s) + 1] = t.c; }
    if (scq1 >= 0) { const SinCos t = sincos_cw(qv[scq1]); sc[2 * (18 + s)] = t.s; sc[2 * (18 + s) + 1] = t.c; }
    // root free-flyer (joint 1): R from the quaternion exactly as Eigen's toRotationMatrix, p = xyz; R column-major then p
    if (s == 0) {
      double Rt[9];
      quat_to_R(qv + 3, Rt);
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) oMi[12 + 3 * c + rr] = Rt[3 * rr + c];
      oMi[12 + 9] = qv[0]; oMi[12 + 10] = qv[1]; oMi[12 + 11] = qv[2];
    }
  }
    WSYNC();
#pragma unroll 1
  for (int L = 0; L < 5; ++L) {   // (second pass)
    const DevPlan::PkJoint fk = fkn;
    if (L + 1 < 5) fkn = P.pk_fk[L + 1][s];          // next level's record is on its way while this level is computed
    const int j = fk.joint;
    if (j >= 0) {
      const bool rev = fk.rev != 0;
      const int a0 = fk.a0, a1 = fk.a1, a2 = fk.a2;
      const double* Pp = oMi + 12 * fk.parent;
      const double sn = rev ? sc[2 * j] : 0.0, cs = rev ? sc[2 * j + 1] : 1.0;
      const double pris = rev ? 0.0 : qv[fk.q_idx];
      double Av[3], Bv[3], Cv[3], Pv[3];
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) { Av[rr] = Pp[a0 + rr]; Bv[rr] = Pp[a1 + rr]; Cv[rr] = Pp[a2 + rr]; Pv[rr] = Pp[9 + rr]; }
      double* Po = oMi + 12 * j;
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        Po[a0 + rr] = Av[rr];
        Po[a1 + rr] = cs * Bv[rr] + sn * Cv[rr];
        Po[a2 + rr] = cs * Cv[rr] - sn * Bv[rr];
        Po[9 + rr] = Pv[rr] + Av[rr] * (fk.t0 + pris) + Bv[rr] * fk.t1 + Cv[rr] * fk.t2;
      }
    }
    WSYNC();
  }
    lin0[0] = lin0[1] = lin0[2] = 0.0; ang0[0] = ang0[1] = ang0[2] = 0.0; lin1[0] = lin1[1] = lin1[2] = 0.0;
    if (s < n) {
      const double* Pj = oMi + 12 * c0_joint;
      const double pj[3] = {Pj[9], Pj[10], Pj[11]};
      if (c0_ang >= 0) { ang0[0] = Pj[3 * c0_ang]; ang0[1] = Pj[3 * c0_ang + 1]; ang0[2] = Pj[3 * c0_ang + 2]; cross3(pj, ang0, lin0); }
      if (c0_lin >= 0) { lin0[0] = Pj[3 * c0_lin]; lin0[1] = Pj[3 * c0_lin + 1]; lin0[2] = Pj[3 * c0_lin + 2]; }
    }
    if (s < nl) {
      const double* Pj = oMi + 12 * c1_joint;
      const double pj[3] = {Pj[9], Pj[10], Pj[11]};
      if (c1_ang >= 0) { const double a1[3] = {Pj[3 * c1_ang], Pj[3 * c1_ang + 1], Pj[3 * c1_ang + 2]}; cross3(pj, a1, lin1); }
      if (c1_lin >= 0) { lin1[0] = Pj[3 * c1_lin]; lin1[1] = Pj[3 * c1_lin + 1]; lin1[2] = Pj[3 * c1_lin + 2]; }
    }
    {
      const double* Pr = oMi + 12 * M.frame_joint[WBC_FR_TRUNK];
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) Rtr[3 * rr + c] = Pr[3 * c + rr];
      ptr[0] = Pr[9]; ptr[1] = Pr[10]; ptr[2] = Pr[11];
    }
    WSYNC();   // oMi is dead again
  }
  if (QCON) {
    if (s < nl) { Kb[4 * s] = lin1[0]; Kb[4 * s + 1] = lin1[1]; Kb[4 * s + 2] = lin1[2]; }
    if (s < 6) { Bb[4 * s] = lin0[0]; Bb[4 * s + 1] = lin0[1]; Bb[4 * s + 2] = lin0[2]; }
  }

  // ---- constraint rows that stay: trunk box (trunkConstraint, :707-754) on the base columns; bounds on the row's own lane
  double clb = 0.0, cub = 0.0;
  if (c_con_trunk) {
    double wxp[3];
    cross3(ang0, ptr, wxp);
    if (s < 6) { I.Cq[0 * 6 + s] = lin0[2] + wxp[2]; I.Cq[1 * 6 + s] = ang0[0]; I.Cq[2 * 6 + s] = ang0[1]; I.Cq[3 * 6 + s] = ang0[2]; }
    const double ay = (s == 1) ? Rtr[7] : ((s == 2) ? -Rtr[6] : Rtr[3]);
    const double ax = (s == 1) ? Rtr[8] : ((s == 2) ? sqrt(fma(Rtr[7], Rtr[7], Rtr[8] * Rtr[8])) : Rtr[0]);
    const double eul = atan2(ay, ax);        // lanes 1, 2, 3 hold roll, pitch, yaw
    const double* bc = V.in + 34;
    if (s < 4) {
      const double cr = (s == 0) ? ptr[2] : eul;
      const double vr = (s == 0) ? bc[0] * tb_z : tb_a;
      clb = (((bc[s] - vr) - cr) * inv_dt) * tb_s;
      cub = (((bc[s] + vr) - cr) * inv_dt) * tb_s;
    }
  }
  // ---- velDamperJointConstraints (:572-637): of reduced variable s and of leg DoF s
  double lb = 0.0, ub = 0.0, lb1 = 0.0, ub1 = 0.0;
  {
    auto damper = [&](const double qi, const double lo, const double hi, const double vm, double& l_, double& u_) {
      if (qi <= lo + dqi) { l_ = -dcoef * (qi - lo - dqs) / (dqi - dqs); if (l_ > vm) l_ = vm; if (l_ < -vm) l_ = -vm; } else l_ = -vm;
      if (qi >= hi - dqi) { u_ = dcoef * (hi - qi - dqs) / (dqi - dqs); if (u_ < -vm) u_ = -vm; if (u_ > vm) u_ = vm; } else u_ = vm;
      if (l_ > 0) l_ = -l_;
      if (u_ < 0) u_ = -u_;
    };
    if (s < n) damper(qv[dq0], dlo0, dhi0, dvm0, lb, ub);
    if (s < nl) damper(qv[dq1], dlo1, dhi1, dvm1, lb1, ub1);
  }
  WSYNC();

  // ---- G_e = -K_e^-1 B_e: lane l = 3 f + i owns row i of foot f (the base block B is the same for every foot)
  unsigned fmask = 0;                      // per instance: bit f = foot f's leg block K_f is (numerically) rank deficient
  {
    double grow[6] = {0, 0, 0, 0, 0, 0};
    const int f = (s < nl) ? s / 3 : 0, i = (s < nl) ? s - 3 * f : 0;
    const double* k0 = Kb + 4 * (3 * f); const double* k1 = k0 + 4; const double* k2 = k1 + 4;   // columns of K_f (leg DoF 0, 1, 2 of the foot)
    const double k00 = k0[0], k10 = k0[1], k20 = k0[2], k01 = k1[0], k11 = k1[1], k21 = k1[2], k02 = k2[0], k12 = k2[1], k22 = k2[2];
    const double a00 = k11 * k22 - k12 * k21, a01 = k02 * k21 - k01 * k22, a02 = k01 * k12 - k02 * k11;
    const double a10 = k12 * k20 - k10 * k22, a11 = k00 * k22 - k02 * k20, a12 = k02 * k10 - k00 * k12;
    const double a20 = k10 * k21 - k11 * k20, a21 = k01 * k20 - k00 * k21, a22 = k00 * k11 - k01 * k10;
    const double det = k00 * a00 + k01 * a10 + k02 * a20;
    const double sc_ = fabs(k00) + fabs(k01) + fabs(k02) + fabs(k10) + fabs(k11) + fabs(k12) + fabs(k20) + fabs(k21) + fabs(k22);
    const bool bad = (s < nl) && !(fabs(det) > A.sing_tol * sc_ * sc_ * sc_);
    const unsigned rowbits = (unsigned)((__ballot(bad) >> rbase) & 0xFFFull);
    fmask = ((rowbits & 0x7u) ? 1u : 0u) | ((rowbits & 0x38u) ? 2u : 0u) | ((rowbits & 0x1C0u) ? 4u : 0u) | ((rowbits & 0xE00u) ? 8u : 0u);
    const double id = -1.0 / det;
    const double r0 = (i == 0) ? a00 : (i == 1) ? a10 : a20, r1 = (i == 0) ? a01 : (i == 1) ? a11 : a21, r2 = (i == 0) ? a02 : (i == 1) ? a12 : a22;
    if (s < nl) {
#pragma unroll
      for (int c = 0; c < 6; ++c) grow[c] = id * (r0 * Bb[4 * c] + r1 * Bb[4 * c + 1] + r2 * Bb[4 * c + 2]);
    }
    if (s < 12) V.xv[s] = g1;
    // leg-bound rows: row p_keep + l = G_l with the leg DoF's velocity bounds; the bounds move p_keep lanes up through LDS
    if (s < nl) {
#pragma unroll
      for (int c = 0; c < 6; c += 2) sts2(I.Cq + (p_keep + s) * 6 + c, grow[c], grow[c + 1]);
      V.cl[p_keep + s] = lb1; V.cl[16 + p_keep + s] = ub1;
    }
  }
  if (s < p_keep) { V.cl[s] = clb; V.cl[16 + s] = cub; }
  WSYNC();
  clb = (s < p) ? V.cl[s] : 0.0;
  cub = (s < p) ? V.cl[16 + s] : 0.0;
  // ---- a rank-deficient leg block (rare; whole-wave branch, per-instance predicates): K_f P = Q R by column-pivoted Gram-Schmidt
  // on lane f of the instance (see process_sim3): z0, z1 are eliminated as usual, the third contact row E q̇_base + r22 z2 = 0 and
  // the leg velocity z2 pivoted last are dealt with by the SWAP further down. Ex[f] = E [6], r22, g0x, g1x, l0, l1, l2.
  double* const Ex = I.M1;                        // [4][12]: M1 is free between the FK and the Cholesky sweep (M2 is full: At, Kb, Bb)
  bool defer = false;
  if (__ballot(valid && fmask != 0)) {
    bool bad_rank = false;
    if (s < 4 && ((fmask >> s) & 1u)) {
      const int f = s;
      const double* k0 = Kb + 4 * (3 * f); const double* k1 = k0 + 4; const double* k2 = k1 + 4;
      const double ca[3] = {k0[0], k0[1], k0[2]}, cb[3] = {k1[0], k1[1], k1[2]}, cc[3] = {k2[0], k2[1], k2[2]};
      const double na = ca[0] * ca[0] + ca[1] * ca[1] + ca[2] * ca[2], nb = cb[0] * cb[0] + cb[1] * cb[1] + cb[2] * cb[2],
                   nc = cc[0] * cc[0] + cc[1] * cc[1] + cc[2] * cc[2];
      const int p0 = (na >= nb && na >= nc) ? 0 : ((nb >= nc) ? 1 : 2);
      double u_[3], v_[3], w_[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        u_[i] = (p0 == 0) ? ca[i] : (p0 == 1) ? cb[i] : cc[i];
        v_[i] = (p0 == 0) ? cb[i] : ca[i];
        w_[i] = (p0 == 2) ? cb[i] : cc[i];
      }
      const int iv = (p0 == 0) ? 1 : 0, iw = (p0 == 2) ? 1 : 2;
      const double r00 = sqrt(u_[0] * u_[0] + u_[1] * u_[1] + u_[2] * u_[2]);
      const double q0[3] = {u_[0] / r00, u_[1] / r00, u_[2] / r00};
      const double rv = q0[0] * v_[0] + q0[1] * v_[1] + q0[2] * v_[2], rw = q0[0] * w_[0] + q0[1] * w_[1] + q0[2] * w_[2];
#pragma unroll
      for (int i = 0; i < 3; ++i) { v_[i] = fma(-rv, q0[i], v_[i]); w_[i] = fma(-rw, q0[i], w_[i]); }
      const double nv2 = v_[0] * v_[0] + v_[1] * v_[1] + v_[2] * v_[2], nw2 = w_[0] * w_[0] + w_[1] * w_[1] + w_[2] * w_[2];
      const bool sw = nw2 > nv2;
      const int p1 = sw ? iw : iv, p2 = sw ? iv : iw;
      const double r01 = sw ? rw : rv, r02 = sw ? rv : rw;
      double s1[3], s2[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) { s1[i] = sw ? w_[i] : v_[i]; s2[i] = sw ? v_[i] : w_[i]; }
      const double r11 = sqrt(sw ? nw2 : nv2);
      const double q1[3] = {s1[0] / r11, s1[1] / r11, s1[2] / r11};
      const double r12 = q1[0] * s2[0] + q1[1] * s2[1] + q1[2] * s2[2];
      double q2[3];
      cross3(q0, q1, q2);
      const double r22 = q2[0] * s2[0] + q2[1] * s2[1] + q2[2] * s2[2];
      bad_rank = !(r11 > 1e-9 * r00) || !(r00 > 0.0);
      const int l0 = 3 * f + p0, l1 = 3 * f + p1, l2 = 3 * f + p2;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const double bx = Bb[4 * c], by = Bb[4 * c + 1], bz = Bb[4 * c + 2];
        const double t0 = q0[0] * bx + q0[1] * by + q0[2] * bz, t1 = q1[0] * bx + q1[1] * by + q1[2] * bz,
                     t2 = q2[0] * bx + q2[1] * by + q2[2] * bz;
        const double g1c = -t1 / r11;
        I.Cq[(p_keep + l1) * 6 + c] = g1c;
        I.Cq[(p_keep + l0) * 6 + c] = -(t0 + r01 * g1c) / r00;
        I.Cq[(p_keep + l2) * 6 + c] = 0.0;
        Ex[12 * f + c] = t2;
      }
      const double g1x = -r12 / r11;
      Ex[12 * f + 6] = r22; Ex[12 * f + 7] = -(r01 * g1x + r02) / r00; Ex[12 * f + 8] = g1x;
      Ex[12 * f + 9] = (double)l0; Ex[12 * f + 10] = (double)l1; Ex[12 * f + 11] = (double)l2;
    }
    defer = ((__ballot(bad_rank) >> rbase) & 0xFFFFull) != 0;
    if (A.dbg_force_defer) defer = fmask != 0;   // diagnostic: every flagged instance takes the tail's general path instead of the swap
    WSYNC();
  }
  // g' = Z'g and H' += d^2 G'G on the base block
  if (s < 6) {
    const double d2 = dpost * dpost;
    double gg[6] = {0, 0, 0, 0, 0, 0}, gs = 0.0;
    const double* const Gr = I.Cq + p_keep * 6;   // G rows (rows beyond nl: never written here -> must not be read)
#pragma unroll
    for (int l = 0; l < 12; ++l) {
      if (l >= nl) break;
      const double gl = Gr[l * 6 + s];
      const double2a t0 = lds2(Gr + l * 6), t1 = lds2(Gr + l * 6 + 2), t2 = lds2(Gr + l * 6 + 4);
      gg[0] = fma(gl, t0.x, gg[0]); gg[1] = fma(gl, t0.y, gg[1]); gg[2] = fma(gl, t1.x, gg[2]);
      gg[3] = fma(gl, t1.y, gg[3]); gg[4] = fma(gl, t2.x, gg[4]); gg[5] = fma(gl, t2.y, gg[5]);
      gs = fma(gl, V.xv[l], gs);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) h[c] = fma(d2, gg[c], h[c]);
    g += gs;
  }
  // ---- the SWAP: for a pivoted foot the kept row E y_base + r22 z2 = 0 is solved for the base unknown with the largest
  // coefficient, y_c* = u'(y, z2), and z2 takes that unknown's slot. The reduced problem keeps its size (n' unknowns, the same 16
  // rows) however many feet are pivoted, nothing is divided by r22, and the pivot |E_c*| >= 0.4 |E| (the first three columns
  // of Q'B are rows of a rotation). In the new unknowns: H'' = M'H_ext M, g'' = M'g_ext, every row C_r <- C_r + C_rc* u' (+ its z2
  // coefficient in slot c*), the bound row of leg DoF l2 becomes the bound row of base DoF c* and vice versa; at the end lane c*
  // delivers z2 = q̇ of leg DoF l2 and row lane l2 delivers q̇ of base DoF c*.
  int dofA = dof0, dofB = dof1;            // DoF whose velocity lane s delivers as reduced variable / as eliminated-leg row
  if (__ballot(valid && fmask != 0 && !defer)) {
#pragma unroll 1
    for (int f = 0; f < 4; ++f) {
      const bool on = valid && !defer && ((fmask >> f) & 1u);
      if (!__ballot(on)) continue;
      const double d2 = dpost * dpost;
      const double Ec = (on && s < 6) ? Ex[12 * f + s] : 0.0;
      // (any of the six slots may be taken, also one that an earlier swap already gave to a leg velocity: the base parts of the
      //  kept rows all lie in the 3-dimensional row space of B, so a fourth pivoted foot finds its pivot only there)
      const double cand = (on && s < 6) ? fabs(Ec) : -1.0;
      const double emax = -rmin16(-cand);
      const int cstar_ = __ffs((int)((__ballot(cand == emax && cand >= 0.0) >> rbase) & 0xFFFFull)) - 1;
      const int cstar = cstar_ < 0 ? 0 : cstar_;
      const double Ecs = bperm(Ec, rbase + cstar);
      const double r22 = Ex[12 * f + 6], g0x = Ex[12 * f + 7], g1x = Ex[12 * f + 8];
      const int l0 = on ? (int)Ex[12 * f + 9] : 0, l1 = on ? (int)Ex[12 * f + 10] : 0, l2 = on ? (int)Ex[12 * f + 11] : 0;
      const double wz = -r22 / Ecs;
      const double us = (s < 6) ? ((s == cstar) ? wz : -Ec / Ecs) : 0.0;
      const double hyz = (s < 6) ? d2 * (I.Cq[(p_keep + l0) * 6 + s] * g0x + I.Cq[(p_keep + l1) * 6 + s] * g1x) : 0.0;
      const double hzz = d2 * (g0x * g0x + g1x * g1x + 1.0);
      const double gz = g0x * V.xv[l0] + g1x * V.xv[l1] + V.xv[l2];
      WSYNC();
      if (on) {
        V.yv[s] = us; V.dv[s] = hyz;
        if (s == cstar) {
#pragma unroll
          for (int k = 0; k < PV; ++k) V.tv[k] = h[k];
        }
      }
      WSYNC();
      const double gcs = bperm(g, rbase + cstar);
      if (on && s < n) {
        const double Hcc = V.tv[cstar], hyzc = V.dv[cstar];
        double hic = 0.0;
#pragma unroll
        for (int j = 0; j < PV; ++j) hic = (j == cstar) ? h[j] : hic;
        const double kz = wz * Hcc + hyzc;
        if (s != cstar) {
#pragma unroll
          for (int j = 0; j < PV; ++j) {
            const double uj = V.yv[j], Hcj = V.tv[j];
            h[j] = (j == cstar) ? wz * hic + hyz + us * kz : h[j] + us * Hcj + hic * uj + us * uj * Hcc;
          }
          g = fma(us, gcs, g);
        } else {
#pragma unroll
          for (int j = 0; j < PV; ++j) {
            const double uj = V.yv[j], Hcj = V.tv[j], hj = V.dv[j];
            h[j] = (j == cstar) ? wz * wz * Hcc + 2.0 * wz * hyzc + hzz : wz * Hcj + hj + uj * kz;
          }
          g = wz * gcs + gz;
        }
      }
      // rows, in place (each lane its own row); row p_keep + l2 becomes the expression of base DoF c*
      if (on && s < p) {
        double* row = I.Cq + s * 6;
        const double crc = row[cstar];
        const double crz = (s == p_keep + l0) ? g0x : ((s == p_keep + l1) ? g1x : 0.0);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const double uc = V.yv[c];
          row[c] = (s == p_keep + l2) ? uc : ((c == cstar) ? crc * wz + crz : row[c] + crc * uc);
        }
      }
      // the kept rows of the feet still to come, in the new unknowns
#pragma unroll 1
      for (int f2 = f + 1; f2 < 4; ++f2) {
        const double e2 = Ex[12 * f2 + cstar];
        WSYNC();
        if (on && s < 6) Ex[12 * f2 + s] = (s == cstar) ? e2 * wz : Ex[12 * f2 + s] + e2 * V.yv[s];
      }
      // bounds: slot c* is leg DoF l2 now, row p_keep + l2 is base DoF c*; and what the two lanes deliver at the end
      {
        const int rl2 = rbase + ((p_keep + l2) & 15);
        const double nl_ = bperm(clb, rl2), nu_ = bperm(cub, rl2), ol_ = bperm(lb, rbase + cstar), ou_ = bperm(ub, rbase + cstar);
        const int dB = bpermi(dofB, rbase + (l2 & 15)), dA = bpermi(dofA, rbase + cstar);
        if (on && s == cstar) { lb = nl_; ub = nu_; dofA = dB; }
        if (on && s == p_keep + l2) { clb = ol_; cub = ou_; }
        if (on && s == l2) dofB = dA;
      }
      WSYNC();
    }
  }
  // a row of the batch tail does nothing; an instance with a leg block of rank < 2 is left to this kernel's tail (general path, same wave)
  PSTOP(3, h[0] + h[3] + g + clb + cub + lb + ub);
  const bool flagged = defer;
  bool live = valid && !flagged;
  if (valid && flagged && s == 0 && A.defer_stat) {   // statistic "deferred_last": (launch sequence number, count) in one word, no reset launch
    unsigned long long old = *(volatile unsigned long long*)A.defer_stat, assumed;
    do {
      assumed = old;
      const unsigned long long cnt = ((assumed >> 32) == (unsigned long long)A.tick_seq) ? (assumed & 0xFFFFFFFFull) + 1ull : 1ull;
      old = atomicCAS(A.defer_stat, assumed, ((unsigned long long)A.tick_seq << 32) | cnt);
    } while (old != assumed);
  }
  WSYNC();

  // ================================ the QP, four at a time =========================================
  // (qp_core's method; no equalities and no fixed variables are left in this problem, so slot 0 is the first inequality slot)
  const bool has_b = s < n, has_r = s < p;
  int status = WBC_QP_OPTIMAL;
  if (live && ((has_b && ((lb != lb) || (ub != ub))) || (has_r && ((clb != clb) || (cub != cub))))) status = WBC_QP_NUMERICAL;
  {
    const unsigned long long nb = __ballot(status != WBC_QP_OPTIMAL);
    if ((nb >> rbase) & 0xFFFFull) { status = WBC_QP_NUMERICAL; live = false; }
  }
  // ---- Cholesky H' = L L' fused with the substitution L y = e_s (rotating registers; column broadcast through V.cl)
  WSYNC();
  V.cl[s] = 0.0; V.cl[16 + s] = 0.0;        // (the row bounds were staged there)
  V.yv[s] = 0.0; V.tv[s] = 0.0;             // second column vector of the blocked sweep: yv | tv, 32 contiguous entries, zero tail
  double y[PV];
#pragma unroll
  for (int k = 0; k < PV; ++k) y[k] = (k == s) ? 1.0 : 0.0;
  double pmin = 1.0;
  // Two columns per trip: the raw columns j and j + 1 of every row go through LDS together and each lane redoes, for the rows below, the
  // one update that column j + 1 receives from step j — the same operations in the same order as two single steps (bit-identical), one LDS
  // round trip instead of two in the 12-step chain.
#pragma unroll 1
  for (int j = 0; j < PV; j += 2) {
    WSYNC();
    if (s < PV) { V.cl[s] = h[0]; V.yv[s] = h[1]; }
    WSYNC();
    const double* c0 = V.cl + j;
    const double* c1 = V.yv + j;
    double cm0[PV], cm1[PV];
#pragma unroll
    for (int rr = 0; rr < PV; rr += 2) {     // (j is even: the columns come in 16-byte pairs, half the LDS instructions of entry-wise reads)
      const double2a v0 = lds2(c0 + rr), v1 = lds2(c1 + rr);
      cm0[rr] = v0.x; cm0[rr + 1] = v0.y; cm1[rr] = v1.x; cm1[rr + 1] = v1.y;
    }
    const double pj = cm0[0];
    pmin = (pj > 0.0) ? fmin(pmin, pj) : -1.0;
    const double rinv = rsqrt(pj), ipj = rinv * rinv;
    // step j on this row
    const double th = h[0] * ipj, ty = y[0] * ipj, yk = y[0] * rinv;
    const double h1 = fma(-th, cm0[1], h[1]), y1 = fma(-ty, cm0[1], y[1]);
    // step j as it acts on column j + 1 of the rows below (what their own lanes compute for themselves)
    const double a = cm0[1];
#pragma unroll
    for (int rr = 1; rr < PV; ++rr) cm1[rr] = fma(-(cm0[rr] * ipj), a, cm1[rr]);
    const double pj2 = cm1[1];
    pmin = (pj2 > 0.0) ? fmin(pmin, pj2) : -1.0;
    const double rinv2 = rsqrt(pj2), ipj2 = rinv2 * rinv2;
    const double th2 = h1 * ipj2, ty2 = y1 * ipj2, yk2 = y1 * rinv2;
#pragma unroll
    for (int rr = 2; rr < PV; ++rr) h[rr - 2] = fma(-th2, cm1[rr], fma(-th, cm0[rr], h[rr]));
#pragma unroll
    for (int rr = 2; rr < PV; ++rr) y[rr - 2] = fma(-ty2, cm1[rr], fma(-ty, cm0[rr], y[rr]));
    y[PV - 2] = fma(-ty2, 0.0, yk); y[PV - 1] = yk2;
    h[PV - 2] = 0.0; h[PV - 1] = 0.0;
  }
  if (live && !(pmin > 0.0)) { status = WBC_QP_NUMERICAL; live = false; }
  PSTOP(4, y[0] + y[11] + h[0]);
  // y = row s of J0 = L^-T.  jf2 = |J0|_F^2 per instance
  double sq = 0.0;
#pragma unroll
  for (int k = 0; k < PV; ++k) sq = fma(y[k], y[k], sq);
  const double jf2 = rsum16(s < PV ? sq : 0.0);
  double* const J = I.M2;
  double* const T = I.M1;
  WSYNC();
  if (s < PV) {
#pragma unroll
    for (int k = 0; k < PV; k += 2) { sts2(J + s * PLD + k, y[k], y[k + 1]); sts2(T + s * PLD + k, 0.0, 0.0); }
  }
  V.tv[s] = g;
  // |C_r|^2 of row s
  double cn2 = 0.0;
  if (has_r) {
    const double2a t0 = lds2(I.Cq + s * 6), t1 = lds2(I.Cq + s * 6 + 2), t2 = lds2(I.Cq + s * 6 + 4);
    cn2 = t0.x * t0.x + t0.y * t0.y + t1.x * t1.x + t1.y * t1.y + t2.x * t2.x + t2.y * t2.y;
  }
  WSYNC();
  // x0 = -J0 (J0' g): the unconstrained minimiser
  double x;
  {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < PV; ++i) t = fma(J[i * PLD + (s < PV ? s : 0)], V.tv[i], t);
    V.dv[s] = has_b ? -t : 0.0;
    WSYNC();
    double xa = 0.0, xb = 0.0;
#pragma unroll
    for (int k = 0; k < PV; k += 2) { const double2a v2 = lds2(V.dv + k); xa = fma(y[k], v2.x, xa); xb = fma(y[k + 1], v2.y, xb); }
    x = has_b ? xa + xb : 0.0;
  }
  PSTOP(5, x + cn2 + jf2);
  // ---- dual active-set iterations (per-row state; loops run until every row of the wave is done)
  int actm = 0;                             // bit 0: this lane's bound is in the working set, bit 1: its row (ONE register: as two bools assigned under
                                            // selected conditions they lived in scratch — two byte loads in every violation scan, a store per change)
  double u = 0.0;
  int a_code = 0, q = 0, iters = 0;
  const int max_iter = 10 * (n + p) + 20;
  bool searching = live;                    // row still iterating
  // d = J'n of constraint (is_row ? row rr_ : bound of variable ip) with sign sgn, on lane s = slot s
  auto normal_d = [&](const bool is_row, const int rr_, const int ip, const double sgn) -> double {
    double d;
    if (is_row) {
      const double2a c0 = lds2(I.Cq + rr_ * 6), c1 = lds2(I.Cq + rr_ * 6 + 2), c2 = lds2(I.Cq + rr_ * 6 + 4);
      d = fma(J[0 * PLD + s], c0.x, fma(J[1 * PLD + s], c0.y, fma(J[2 * PLD + s], c1.x, fma(J[3 * PLD + s], c1.y,
          fma(J[4 * PLD + s], c2.x, J[5 * PLD + s] * c2.y))))) * sgn;
    } else d = sgn * J[(ip & 15) * PLD + s];
    return d;
  };
  // drop slot l of the rows `dr`: Givens sequence read off the removed row of T (rare path)
  auto drop_slot = [&](const bool dr, const int l_) {
    const int l = dr ? l_ : 0;
    const int lc = bpermi(a_code, rbase + l) & 255;
    if (dr && s == ((lc >= n) ? lc - n : lc)) actm &= (lc >= n) ? ~2 : ~1;
    WSYNC();
    V.yv[s] = u; V.tv[s] = (double)a_code;
    WSYNC();
    if (dr && s >= l && s < q - 1) { u = V.yv[s + 1]; a_code = (int)V.tv[s + 1]; }
    if (dr && s == q - 1) { u = 0.0; a_code = 0; }
    const int sv = s < PV ? s : PV - 1;   // (lanes beyond the variables shadow the last row; they never write)
    const int srow = (sv >= l) ? ((sv + 1 < PV) ? sv + 1 : sv) : sv;
    double tx = T[srow * PLD + l];
    double jx = J[sv * PLD + l];
    double hrun = T[l * PLD + l];
    const int kend = dr ? q - 1 : 0;    // this row's rotations run k = l .. q - 2
#pragma unroll 1
    for (int k0 = 0; k0 < PV - 1; ++k0) {
      const bool on = dr && (l + k0 < kend);
      if (!__ballot(on)) break;
      const int k = on ? l + k0 : 0;
      const double tb = T[l * PLD + k + 1];
      const double nrm2 = fma(hrun, hrun, tb * tb);
      double c_ = 1.0, s_ = 0.0, rho = 0.0;
      if (nrm2 > 0.0) { const double ri = rsqrt(nrm2); c_ = tb * ri; s_ = -hrun * ri; rho = nrm2 * ri; }
      const double ty_ = T[srow * PLD + k + 1];
      const double jy = J[sv * PLD + k + 1];
      WSYNC();
      if (on) {
        hrun = rho;
        if (s < q - 1) T[s * PLD + k] = fma(c_, tx, s_ * ty_);
        if (s < n) J[s * PLD + k] = fma(c_, jx, s_ * jy);
        tx = fma(-s_, tx, c_ * ty_);
        jx = fma(-s_, jx, c_ * jy);
      }
      WSYNC();
    }
    WSYNC();
    if (dr) {
      if (s < q) T[s * PLD + q - 1] = 0.0;
    }
    WSYNC();
    if (dr) {
      if (s < q) T[(q - 1) * PLD + s] = 0.0;
      if (s < n) J[s * PLD + q - 1] = jx;
      --q;
    }
    WSYNC();
  };
  // with d staged (V.dv = d, V.yv = d restricted to the slots >= q): z = J2 d2, r = T d1, and the add step's dq = d_q, jq = J[s][q]
  struct Zr { double z, rv, dq, jq; };
  auto products = [&](const bool want_r) -> Zr {
    Zr o;
    double z = 0.0, zb = 0.0, rv = 0.0, rvb = 0.0;
    const int srd = s < PV ? s : PV - 1;
    o.dq = V.dv[q & 15];                                                // d of slot q and this row's J entry there (the add step's): read
    o.jq = J[srd * PLD + (q & 15)];                                     // in the same round as the products below
#pragma unroll
    for (int k = 0; k < PV; k += 2) {
      const double2a j2 = lds2(J + srd * PLD + k); const double2a y2 = lds2(V.yv + k);
      z = fma(j2.x, y2.x, z); zb = fma(j2.y, y2.y, zb);
    }
    z += zb;
    if (want_r) {                           // r = T d1: nothing to do while no row of the wave holds an active inequality
#pragma unroll
      for (int k = 0; k < PV; k += 2) {
        const double2a t2 = lds2(T + srd * PLD + k); const double2a d2 = lds2(V.dv + k);
        rv = fma(t2.x, d2.x, rv); rvb = fma(t2.y, d2.y, rvb);
      }
      rv += rvb;
    }
    if (s >= q) rv = 0.0;
    if (!has_b) z = 0.0;
    o.z = z; o.rv = rv;
    return o;
  };
  // add: Householder P with P d2 = delta e1; J2 <- J2 P; T gets column (-r/delta, 1/delta); the new slot's multiplier is u_new
  auto add_step = [&](const bool add, const double zn, const Zr& zr, const int wc, const bool is_row, const int rr_, const int ip, const double u_new) {
    const double rsz = frsq(zn), sz = zn * rsz;
    const double delta = (zr.dq >= 0.0) ? -sz : sz;
    const double hv = zn - delta * zr.dq;               // v'v / 2
    const double vv = 2.0 * hv;
    const double w = (zr.z - delta * zr.jq) * ((vv > 0.0) ? frcp(hv) : 0.0);
    if (add && has_b && vv > 0.0) {
      // J2 <- J2 - w v', v = d2 - delta e_q: the sweep runs on d2 alone (yv = d for k >= q, else 0) and entry q is then stored with its own term —
      // the same arithmetic as selecting v_k inside the loop, without two compares and four selects per pair
#pragma unroll
      for (int k = 0; k < PV; k += 2) {
        const double2a j2 = lds2(J + s * PLD + k); const double2a y2 = lds2(V.yv + k);
        sts2(J + s * PLD + k, fma(-w, y2.x, j2.x), fma(-w, y2.y, j2.y));
      }
      J[s * PLD + q] = fma(-w, zr.dq - delta, zr.jq);
    }
    if (add) {
      const double idel = (zr.dq >= 0.0) ? -rsz : rsz;
      if (s < q) T[s * PLD + q] = -zr.rv * idel;
      if (s == q) { T[s * PLD + q] = idel; u = u_new; a_code = wc; }
      if (s == (is_row ? rr_ : (ip & 15))) actm |= is_row ? 2 : 1;
      ++q;
    }
  };

  // ================================ warm start ======================================================
  int dofR = 0;                             // DoF whose velocity bound row s (>= p_keep) carries (the working set's indexing)
  if (WARM) {
    const unsigned long long ws0 = (unsigned long long)__double_as_longlong(V.in[38]), ws1 = (unsigned long long)__double_as_longlong(V.in[39]);
    dofR = bpermi(dofB, rbase + ((s - p_keep) & 15));
    auto bits = [](const unsigned long long w, const int i) -> int { return (int)(((w >> (i & 31)) & 1ull) | (((w >> (32 + (i & 31))) & 1ull) << 1)); };
    // the seeds seen from the reduced problem: bound of reduced variable s = velocity bound of DoF dofA; row s < p_keep = original
    // constraint row s (the trunk box leads findConstraints' order here); row s >= p_keep = velocity bound of leg DoF dofR
    int sb = has_b ? bits(ws0, dofA) : 0;
    int sr = has_r ? ((s < p_keep) ? bits(ws1, s) : bits(ws0, dofR)) : 0;
    if (sb == 3) sb = 0;
    if (sr == 3) sr = 0;
    // a seed is taken only if the unconstrained minimiser x0 violates it or comes close to it (qp_core, solve_v3 `far`)
    const double x0r = x;
    WSYNC();
    V.xv[s] = x;
    WSYNC();
    const double near = 0.25 * fmax(1.0, -rmin16(has_b ? -fabs(x) : 0.0));
    double vr = 0.0;
    if (has_r) {
      const double2a c0 = lds2(I.Cq + s * 6), c1 = lds2(I.Cq + s * 6 + 2), c2 = lds2(I.Cq + s * 6 + 4);
      const double2a x0 = lds2(V.xv), x1 = lds2(V.xv + 2), x2 = lds2(V.xv + 4);
      vr = fma(c0.x, x0.x, fma(c0.y, x0.y, fma(c1.x, x1.x, fma(c1.y, x1.y, fma(c2.x, x2.x, c2.y * x2.y)))));
    }
    const double slb = (sb == 2) ? ub - x : x - lb;      // slack of the seeded side at x0
    const double slr = (sr == 2) ? cub - vr : vr - clb;
    bool pend_b = live && has_b && ((sb == 1 && lb > -QP_INF) || (sb == 2 && ub < QP_INF)) && (slb <= near);
    bool pend_r = live && has_r && ((sr == 1 && clb > -QP_INF) || (sr == 2 && cub < QP_INF)) && (slr <= near);
    bool seeded = false;
#pragma unroll 1
    for (;;) {                              // one seed per row and pass: bounds first, then rows, lowest index first
      const unsigned mb = (unsigned)((__ballot(pend_b) >> rbase) & 0xFFFFull), mr = (unsigned)((__ballot(pend_r) >> rbase) & 0xFFFFull);
      const bool seeding = (mb | mr) != 0u;
      if (!__ballot(seeding)) break;
      const bool is_row = mb == 0u;
      const int idx = seeding ? __ffs((int)(is_row ? mr : mb)) - 1 : 0;
      if (seeding && s == idx) { if (is_row) pend_r = false; else pend_b = false; }
      const int c_side = ((is_row ? sr : sb) == 2) ? 256 : 0;
      const double c_n2 = is_row ? cn2 : 1.0;
      const int wsrc = rbase + idx;
      const int wc = ((is_row ? n + idx : idx) & 255) | bpermi(c_side, wsrc);
      const double np2 = bperm(c_n2, wsrc);
      const int ip = wc & 255;
      const int rr_ = is_row ? ip - n : 0;
      const double sgn = (wc >> 8) ? -1.0 : 1.0;
      double d = normal_d(is_row, rr_, ip, sgn);
      if (!has_b || !seeding) d = 0.0;
      WSYNC();
      V.dv[s] = d; V.yv[s] = (s >= q) ? d : 0.0;
      WSYNC();
      const double zn = rsum16(s >= q ? d * d : 0.0);
      const Zr zr = products(__ballot(seeding && q > 0) != 0);
      const bool add = seeding && (zn > 100.0 * n * EPS2 * jf2 * np2);      // (a dependent seed is simply not taken)
      if (__ballot(add)) {
        add_step(add, zn, zr, wc, is_row, rr_, ip, 0.0);
        if (add) { seeded = true; ++iters; }
      }
    }
    // x, u from the factors: with s_j = b_j - n_j'x0 the slacks of the slots at x0:  w = T's,  x = x0 + J1 w,  u = T w
    auto refresh = [&](const bool on) {
      const int cc = a_code & 255;
      const double sb_ = bperm(-slb, rbase + (cc & 15)), sr_ = bperm(-slr, rbase + ((cc - n) & 15));
      const double sj = (s < q) ? ((cc < n) ? sb_ : sr_) : 0.0;
      WSYNC();
      V.dv[s] = sj;
      WSYNC();
      const int sv = s < PV ? s : PV - 1;
      double w = 0.0;
#pragma unroll
      for (int j = 0; j < PV; ++j) w = fma(T[j * PLD + sv], V.dv[j], w);        // column s of T (zero outside the slots)
      WSYNC();
      V.yv[s] = (s < q && s < PV) ? w : 0.0;
      WSYNC();
      double xa = 0.0, ua = 0.0;
#pragma unroll
      for (int k = 0; k < PV; k += 2) {
        const double2a j2 = lds2(J + sv * PLD + k), t2 = lds2(T + sv * PLD + k), w2 = lds2(V.yv + k);
        xa = fma(j2.x, w2.x, fma(j2.y, w2.y, xa)); ua = fma(t2.x, w2.x, fma(t2.y, w2.y, ua));
      }
      if (on) { x = has_b ? x0r + xa : 0.0; u = (s < q) ? ua : 0.0; }
    };
    if (__ballot(seeded)) {
      refresh(seeded);
      // RESTORATION: while a seeded multiplier is negative the most negative slot is dropped and the iterate moved to the minimiser on
      // the remaining set (the add step read backwards: x <- x - u_l z, u <- u + u_l r with z, r of the dropped constraint on the NEW
      // factors); after any drop x, u are rebuilt once more from the factors (they went through where the wrong seeds put them: with
      // cond(H) ~ 1e9 that costs digits; the factors saw orthogonal updates only) and one more pass runs on the accurate multipliers.
      bool restoring = seeded, did = false, again = false;
#pragma unroll 1
      for (;;) {
        const double um = rmin16((s < q) ? u : 0.0);
        bool rest = restoring && (um < 0.0);
        if (rest && ++iters > max_iter) { status = WBC_QP_MAX_ITER; rest = false; restoring = false; searching = false; }
        if (!__ballot(rest)) {
          if (!__ballot(restoring && did && !again)) break;
          const bool on = restoring && did && !again;
          refresh(on);
          if (on) again = true;
          continue;
        }
        const int l = rest ? __ffs((int)((__ballot(rest && s < q && u == um) >> rbase) & 0xFFFFull)) - 1 : 0;
        const int lcode = bpermi(a_code, rbase + (l < 0 ? 0 : l));
        drop_slot(rest, l < 0 ? 0 : l);
        const int ip = lcode & 255;
        const bool is_row = ip >= n;
        const int rr_ = is_row ? ip - n : 0;
        double d = normal_d(is_row, rr_, ip, (lcode >> 8) ? -1.0 : 1.0);
        if (!has_b || !rest) d = 0.0;
        WSYNC();
        V.dv[s] = d; V.yv[s] = (s >= q) ? d : 0.0;
        WSYNC();
        const Zr zr = products(__ballot(rest && q > 0) != 0);
        if (rest) { x = fma(-um, zr.z, x); u = fma(um, zr.rv, u); did = true; }
      }
    }
  }

#pragma unroll 1
  for (;;) {
    // most violated inactive inequality of each row
    WSYNC();
    V.xv[s] = x;
    WSYNC();
    double best = 0.0; int code = -1;
    double cand_b = 0.0, cand_n2 = 1.0;    // bound value (signed by side) and |normal|^2 of this lane's candidate: fetched with its code in ONE round
    if (has_b && !(actm & 1)) {
      if (lb > -QP_INF) { const double sl = x - lb; if (sl < -1e-9 * fmax(1.0, fabs(lb)) && sl < best) { best = sl; code = s; cand_b = lb; } }
      if (ub < QP_INF) { const double sl = ub - x; if (sl < -1e-9 * fmax(1.0, fabs(ub)) && sl < best) { best = sl; code = s | 256; cand_b = -ub; } }
    }
    if (has_r && !(actm & 2)) {
      const double2a c0 = lds2(I.Cq + s * 6), c1 = lds2(I.Cq + s * 6 + 2), c2 = lds2(I.Cq + s * 6 + 4);
      const double2a x0 = lds2(V.xv), x1 = lds2(V.xv + 2), x2 = lds2(V.xv + 4);
      const double v = fma(c0.x, x0.x, fma(c0.y, x0.y, fma(c1.x, x1.x, fma(c1.y, x1.y, fma(c2.x, x2.x, c2.y * x2.y)))));
      if (clb > -QP_INF) { const double sl = v - clb; if (sl < -1e-9 * fmax(1.0, fabs(clb)) && sl < best) { best = sl; code = n + s; cand_b = clb; cand_n2 = cn2; } }
      if (cub < QP_INF) { const double sl = cub - v; if (sl < -1e-9 * fmax(1.0, fabs(cub)) && sl < best) { best = sl; code = (n + s) | 256; cand_b = -cub; cand_n2 = cn2; } }
    }
    const double worst = rmin16(best);
    if (searching && !(worst < 0.0)) searching = false;               // primal feasible -> this row is optimal
#ifdef WBC_ABLATE
    if (A.dbg_stop == 108) searching = false;                         // timing cut: one violation scan, no working-set change
#endif
    if (!__ballot(searching)) break;
    const unsigned long long wm = __ballot(searching && best == worst);
    const int wl = __ffs((int)((wm >> rbase) & 0xFFFFull)) - 1;      // first lane of the row holding the worst violation
    const int wsrc = rbase + (wl < 0 ? 0 : wl);                       // (lane s evaluates bound s and row s: the candidate's data sit on its own lane)
    const int wc = bpermi(code, wsrc);
    const double b_ip = bperm(cand_b, wsrc);
    const double np2 = bperm(cand_n2, wsrc);
    const int ip = wc & 255, ip_side = (wc >> 8) & 1;
    const bool is_row = ip >= n;
    const int rr_ = is_row ? ip - n : 0;
    const double sgn = ip_side ? -1.0 : 1.0;
    double s_ip = worst, u_ip = 0.0;
    bool stepping = searching;              // row inside the partial-step loop for its constraint
    int drop_l = -1;
#pragma unroll 1
    for (;;) {
      if (stepping && ++iters > max_iter) { status = WBC_QP_MAX_ITER; stepping = false; searching = false; }
      // ---- drop slot l of the rows that ask for it
      if (__ballot(stepping && drop_l >= 0)) {
        const bool dr = stepping && drop_l >= 0;
        drop_slot(dr, drop_l);
        // slack of the constraint being added, at the current x
        V.xv[s] = x;
        WSYNC();
        if (dr) {
          double v;
          if (is_row) {
            const double2a c0 = lds2(I.Cq + rr_ * 6), c1 = lds2(I.Cq + rr_ * 6 + 2), c2 = lds2(I.Cq + rr_ * 6 + 4);
            const double2a x0 = lds2(V.xv), x1 = lds2(V.xv + 2), x2 = lds2(V.xv + 4);
            v = fma(c0.x, x0.x, fma(c0.y, x0.y, fma(c1.x, x1.x, fma(c1.y, x1.y, fma(c2.x, x2.x, c2.y * x2.y)))));
          } else v = V.xv[ip & 15];
          s_ip = sgn * v - b_ip;
          drop_l = -1;
        }
      }
      if (!__ballot(stepping)) break;
      // ---- d = J'n, z = J2 d2, r = T d1
      double d = normal_d(is_row, rr_, ip, sgn);
      if (!has_b || !stepping) d = 0.0;      // (lanes >= n read padding: masked here)
      WSYNC();
      V.dv[s] = d; V.yv[s] = (s >= q) ? d : 0.0;
      WSYNC();
      const double zn = rsum16(s >= q ? d * d : 0.0);
      const Zr zr = products(__ballot(stepping && q > 0) != 0);
      const double z = zr.z, rv = zr.rv;
      const bool have_step = zn > 100.0 * n * EPS2 * jf2 * np2;
      const bool cand = (s < q) && (rv > 2.2250738585072014e-308);   // (normal: frcp's estimate of a denormal is inf)
      const double ratio = cand ? u * frcp(rv) : INFINITY;
      const double t1 = rmin16(ratio);
      const unsigned long long lm = __ballot(cand && ratio == t1);
      const int l = (t1 < INFINITY) ? __ffs((int)((lm >> rbase) & 0xFFFFull)) - 1 : -1;
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
        if (add) stepping = false;          // this row goes back to the search
      }
      if (stepping) drop_l = l;             // blocking slot: dropped at the top of the next pass, then the step is retried
    }
  }
  if (status == WBC_QP_OPTIMAL) {
    const unsigned long long bad = __ballot(has_b && !(fabs(x) <= 1.7976931348623157e308));
    if ((bad >> rbase) & 0xFFFFull) status = WBC_QP_NUMERICAL;
  }
  if (status != WBC_QP_OPTIMAL) x = 0.0;
  if (WARM && A.ws_out) {   // the final working set in FULL-problem indexing (KernelArgs.ws_in); an unsolved QP carries nothing
    const int cc = a_code & 255, sd = (a_code >> 8) & 1;
    const int dA = bpermi(dofA, rbase + (cc & 15));            // slot holds the bound of reduced variable cc: its DoF
    const int dR = bpermi(dofR, rbase + ((cc - n) & 15));      // slot holds reduced row cc - n >= p_keep: the leg DoF whose bound it is
    unsigned long long w0 = 0ull, w1 = 0ull;
    if (status == WBC_QP_OPTIMAL && s < q) {
      if (cc < n) w0 = 1ull << (32 * sd + (dA & 31));
      else if (cc - n < p_keep) w1 = 1ull << (32 * sd + ((cc - n) & 31));
      else w0 = 1ull << (32 * sd + (dR & 31));
    }
    w0 = ror16(w0); w1 = ror16(w1);
    if (valid && !flagged && s == 0) { A.ws_out[2 * (size_t)b] = w0; A.ws_out[2 * (size_t)b + 1] = w1; }
  }

  // ---- x = Z y, q̇ by DoF through LDS, outputs
  WSYNC();
  V.xv[s] = has_b ? x : 0.0;
  V.cl[s] = 0.0; V.cl[16 + s] = 0.0;
  WSYNC();
  double x1 = 0.0;
  if (s < nl) {
    const double2a v0 = lds2(V.xv), v1 = lds2(V.xv + 2), v2 = lds2(V.xv + 4);
    const double2a g0 = lds2(I.Cq + (p_keep + s) * 6), g1_ = lds2(I.Cq + (p_keep + s) * 6 + 2), g2 = lds2(I.Cq + (p_keep + s) * 6 + 4);
    x1 = fma(g0.x, v0.x, fma(g0.y, v0.y, fma(g1_.x, v1.x, fma(g1_.y, v1.y, fma(g2.x, v2.x, g2.y * v2.y)))));
    V.cl[dofB] = x1;
  }
  if (s < n) V.cl[dofA] = x;
  WSYNC();
  const bool wr = valid && !flagged;
  if (wr) {
    double* qo = A.out.qdot + (size_t)b * NV;
    qo[s] = V.cl[s];
    if (16 + s < NV) qo[16 + s] = V.cl[16 + s];
    if (s == 0) {
      A.out.status[b] = status;
      if (A.out.iters) A.out.iters[b] = iters + nl + P.nlock;
    }
  }
  // ---- jointVelocitiestoConfig (Robot_Wrapper4.py:440-441)
  if (A.out.q_next) {
    WSYNC();
    V.xv[s] = (s < 6) ? V.cl[s] * dt : 0.0;
    WSYNC();
    double* qn = A.out.q_next + (size_t)b * NQ;
    if (wr) {
      integrate_ff(V, s, qn);
      // 1-DoF joints: q + v dt, DoF by DoF (two per lane; a locked DoF's velocity is 0, the padding of a smaller model stays 0)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int d = 6 + s + 16 * hh;
        if (d < nv) { const int qi = M.col_q[d]; qn[qi] = qv[qi] + V.cl[d] * dt; }
      }
      if (s < NQ - nq) qn[nq + s] = 0.0;
    }
  }
  // ---- the tail: an instance left out above (a stance-leg block of rank < 2 — never seen on the benchmark distribution — or the
  // diagnostic dbg_force_defer) is redone here, by this wave, on the general path (process_instance, one instance per wavefront, LDS
  // shared with the packed layout). No list, no second launch, and a batch that defers everything runs at the general kernel's occupancy.
#ifndef WBC_NO_TAIL   // (A/B variant builds only: make variant VFLAGS=-DWBC_NO_TAIL measures what carrying the tail costs the common path)
  const unsigned long long tailm = __ballot(valid && flagged && s == 0);
  if (tailm) {
#pragma unroll 1
    for (int rr = 0; rr < 4; ++rr) {
      if (!((tailm >> (16 * rr)) & 1ull)) continue;
      tail_instance<WARM, false>(&SU.G, 4 * (int)blockIdx.x + rr, models, cfgs, plans);
    }
  }
#endif
}
template __global__ void wbc_tick_sim3p_kernel<false, false>(const KernelArgs, const DevModel* __restrict__, const WbcConfig* __restrict__, const DevPlan* __restrict__);
template __global__ void wbc_tick_sim3p_kernel<true, false>(const KernelArgs, const DevModel* __restrict__, const WbcConfig* __restrict__, const DevPlan* __restrict__);
template __global__ void wbc_tick_sim3p_kernel<false, true>(const KernelArgs, const DevModel* __restrict__, const WbcConfig* __restrict__, const DevPlan* __restrict__);
template __global__ void wbc_tick_sim3p_kernel<true, true>(const KernelArgs, const DevModel* __restrict__, const WbcConfig* __restrict__, const DevPlan* __restrict__);
template __global__ void wbc_tick_sim3p_kernel<false, false, true>(const KernelArgs, const DevModel* __restrict__, const WbcConfig* __restrict__, const DevPlan* __restrict__);
template __global__ void wbc_tick_sim3p_kernel<true, false, true>(const KernelArgs, const DevModel* __restrict__, const WbcConfig* __restrict__, const DevPlan* __restrict__);

// ------------------------------------------------------------------------------------------------
// wbc_update_kernel for four instances per wavefront (same lane layout and FK records as the packed tick kernel): the one-instance
// kernel spends a whole wave's instruction stream on one 22-joint FK; in a roll-out that was a third of the closed-loop tick.
// Used when every plan of the batch is DevPlan.pk_update_ok (the packed FK schedule reaches every frame the estimator reads).
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) UInst {
  double oMi[24 * 12];
  double q[32];
  double sc[64];
  double pf[6 * 4];                         // feet 0..3, trunk, gripper: world positions
  double ft[16];
};
struct __attribute__((aligned(16))) USmemP { UInst I[4]; };
__global__ void __launch_bounds__(64) wbc_update_packed_kernel(const UpdateArgs A, const DevModel* __restrict__ models,
                                                               const WbcConfig* __restrict__ cfgs, const DevPlan* __restrict__ plans) {
  __shared__ USmemP UP;
  const int lane = threadIdx.x, r = lane >> 4, s = lane & 15;
  UInst& U = UP.I[r];
  const int b_raw = 4 * blockIdx.x + r;
  const bool valid = b_raw < A.B;
  const size_t b = valid ? b_raw : A.B - 1;
  int mid = 0;
  if (A.model_id) { mid = A.model_id[b]; mid = mid < 0 ? 0 : (mid >= A.n_models ? A.n_models - 1 : mid); }
  const DevModel& M = models[mid];
  const WbcConfig& cfg = cfgs[mid];
  const DevPlan& P = plans[mid];
  const int nq = M.nq;
  const bool warm = A.mode == WBC_ROLLOUT_WARMUP;
  // ---- every global read first. config = [current base xyz, base_config (IMU quaternion), new joint angles]  (:388-389)
  double c0, c1 = 0.0;
  if (warm) c0 = A.q_next[b * NQ + s];
  else if (s < 3) c0 = A.q_cur[b * NQ + s];
  else if (s < 7) c0 = A.imu ? A.imu[b * 4 + (s - 3)] : A.q_next[b * NQ + s];
  else c0 = A.q_next[b * NQ + s];
  if (16 + s < nq) c1 = A.q_next[b * NQ + 16 + s];
  const double ft = (s < 12) ? A.foot_targets[b * 15 + s] : 0.0;
  const double eet = (A.ee_target && s < 15) ? A.ee_target[b * 15 + s] : 0.0;
  const double ees = (A.ee_target && A.ee_step && s < 15) ? A.ee_step[b * 15 + s] : 0.0;
  double rref[3] = {0.0, 0.0, 0.0};
  if (A.ee_prev_rot && A.ee_ref_rot) {
#pragma unroll
    for (int h = 0; h < 3; ++h) if (16 * h + s < 45) rref[h] = A.ee_ref_rot[b * 45 + 16 * h + s];
  }
  const double tt = (A.trunk_target && s < 3) ? A.trunk_target[b * 3 + s] : 0.0;
  const double tts = (A.trunk_target && A.trunk_step && s < 3) ? A.trunk_step[b * 3 + s] : 0.0;
  const double ter = (A.trunk_prev_rot && A.trunk_ref_euler && s < 3) ? A.trunk_ref_euler[b * 3 + s] : 0.0;
  int st = 0, stm = 0, it = 0, its = 0;
  if (s == 0) {
    if (A.status_max) { st = A.status[b]; stm = A.status_max[b]; }
    if (A.iters_sum) { it = A.iters[b]; its = A.iters_sum[b]; }
  }
  DevPlan::PkJoint fkn = P.pk_fk[0][s];
  const int scq0 = P.pk_scq[(2 + s) & 31], scq1 = P.pk_scq[(18 + s) & 31];
  // frame of this lane: feet 0..3, trunk, gripper
  const int fr = (s < 4) ? WBC_FR_EE0 + s : ((s == 4) ? WBC_FR_TRUNK : WBC_FR_EE0 + 4);
  const int fjoint = M.frame_joint[fr];
  const double f0 = M.frame_p[fr][0], f1 = M.frame_p[fr][1], f2 = M.frame_p[fr][2];
  const int tjoint = M.frame_joint[WBC_FR_TRUNK];
  U.q[s] = c0; U.q[16 + s] = c1;
  if (s < 12) U.ft[s] = ft;
  WSYNC();
  const double* const qv = U.q;
  double* const oMi = U.oMi;
  if (scq0 >= 0) { const SinCos t = sincos_cw(qv[scq0]); U.sc[2 * (2 + s)] = t.s; U.sc[2 * (2 + s) + 1] = t.c; }
  if (scq1 >= 0) { const SinCos t = sincos_cw(qv[scq1]); U.sc[2 * (18 + s)] = t.s; U.sc[2 * (18 + s) + 1] = t.c; }
  if (s == 0) {
    double Rt[9];
    quat_to_R(qv + 3, Rt);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) oMi[12 + 3 * c + rr] = Rt[3 * rr + c];
    oMi[12 + 9] = qv[0]; oMi[12 + 10] = qv[1]; oMi[12 + 11] = qv[2];
  }
  WSYNC();
#pragma unroll 1
  for (int L = 0; L < 5; ++L) {             // the packed tick kernel's FK, record for record
    const DevPlan::PkJoint fk = fkn;
    if (L + 1 < 5) fkn = P.pk_fk[L + 1][s];
    const int j = fk.joint;
    if (j >= 0) {
      const bool rev = fk.rev != 0;
      const int a0 = fk.a0, a1 = fk.a1, a2 = fk.a2;
      const double* Pp = oMi + 12 * fk.parent;
      const double sn = rev ? U.sc[2 * j] : 0.0, cs = rev ? U.sc[2 * j + 1] : 1.0;
      const double pris = rev ? 0.0 : qv[fk.q_idx];
      double Av[3], Bv[3], Cv[3], Pv[3];
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) { Av[rr] = Pp[a0 + rr]; Bv[rr] = Pp[a1 + rr]; Cv[rr] = Pp[a2 + rr]; Pv[rr] = Pp[9 + rr]; }
      double* Po = oMi + 12 * j;
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        Po[a0 + rr] = Av[rr];
        Po[a1 + rr] = cs * Bv[rr] + sn * Cv[rr];
        Po[a2 + rr] = cs * Cv[rr] - sn * Bv[rr];
        Po[9 + rr] = Pv[rr] + Av[rr] * (fk.t0 + pris) + Bv[rr] * fk.t1 + Cv[rr] * fk.t2;
      }
    }
    WSYNC();
  }
  if (s < 6) {
    const double* Pj = oMi + 12 * fjoint;
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) U.pf[4 * s + rr] = Pj[9 + rr] + Pj[rr] * f0 + Pj[3 + rr] * f1 + Pj[6 + rr] * f2;
  }
  WSYNC();
  // trunkWorldPos: trunk_pos = WPA - WRB . BPA  (:1321-1325), evaluated by every lane of the instance
  const double* Pt = oMi + 12 * tjoint;     // R column-major
  double WPA[3], BPA[3], base[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double t = U.pf[4 * 4 + i];
    WPA[i] = (U.ft[i] + U.ft[3 + i] + U.ft[6 + i] + U.ft[9 + i]) / 4;
    BPA[i] = ((U.pf[i] - t) + (U.pf[4 + i] - t) + (U.pf[8 + i] - t) + (U.pf[12 + i] - t)) / 4;
  }
#pragma unroll
  for (int rr = 0; rr < 3; ++rr) base[rr] = WPA[rr] - (Pt[rr] * BPA[0] + Pt[3 + rr] * BPA[1] + Pt[6 + rr] * BPA[2]);
  if (warm) { base[0] = qv[0]; base[1] = qv[1]; base[2] = qv[2]; }   // no estimator while warming up (running == False, :414)
  if (!valid) return;
  {
    double* qo = A.q_new + b * NQ;
    qo[s] = (s == 0) ? base[0] : (s == 1) ? base[1] : (s == 2) ? base[2] : c0;
    if (16 + s < NQ) qo[16 + s] = (16 + s < nq) ? c1 : 0.0;
  }
  if (A.grip_trace && s < 3) {
    const double d = (s == 0) ? base[0] - qv[0] : (s == 1) ? base[1] - qv[1] : base[2] - qv[2];
    A.grip_trace[b * 3 + s] = U.pf[4 * 5 + s] + d;
  }
  if (s == 0) {
    if (A.status_max && st > stm) A.status_max[b] = st;
    if (A.iters_sum) A.iters_sum[b] = its + it;
  }
  // ---- side effects of qpb() on the reference state, then the targets move on (as wbc_update_kernel)
  if (A.ee_target && s < 15) {
    const int e = s / 3;
    if (cfg.task_ee[e] && A.prev_ee_target) A.prev_ee_target[b * 15 + s] = eet;       // prev_EE_pos[i] = target (:1151)
    if (A.ee_step) A.ee_target[b * 15 + s] = eet + ees;
  }
  if (A.ee_prev_rot && A.ee_ref_rot) {
#pragma unroll
    for (int h = 0; h < 3; ++h) {
      const int i = 16 * h + s;
      if (i < 45 && cfg.task_ee[i / 9]) A.ee_prev_rot[b * 45 + i] = rref[h];          // prev_EE_CoM_rot[i] = R* (:1152)
    }
  }
  if (A.trunk_target && s < 3) {
    if (cfg.task_trunk && A.prev_trunk_target) A.prev_trunk_target[b * 3 + s] = tt;   // prev_trunk_ref = target (:995)
    if (A.trunk_step) A.trunk_target[b * 3 + s] = tt + tts;
  }
  if (A.trunk_prev_rot && A.trunk_ref_euler && __ballot(cfg.task_trunk != 0)) {      // old_ref_trunk_rot_matrix = R* (:996), with the trunk task on
    const SinCos t = sincos_cw(ter);          // lanes 0..2 of the row: roll, pitch, yaw of the reference
    const int rb = lane & 48;
    const double sa = bperm(t.s, rb), ca = bperm(t.c, rb), sb = bperm(t.s, rb + 1), cb = bperm(t.c, rb + 1), sc_ = bperm(t.s, rb + 2), cc = bperm(t.c, rb + 2);
    double Rs[9];
    Rs[0] = cc * cb; Rs[1] = cc * sb * sa - sc_ * ca; Rs[2] = cc * sb * ca + sc_ * sa;
    Rs[3] = sc_ * cb; Rs[4] = sc_ * sb * sa + cc * ca; Rs[5] = sc_ * sb * ca - cc * sa;
    Rs[6] = -sb;      Rs[7] = cb * sa;                 Rs[8] = cb * ca;
    if (valid && cfg.task_trunk && s < 9) {
      double v = Rs[0];
#pragma unroll
      for (int i = 1; i < 9; ++i) v = (s == i) ? Rs[i] : v;
      A.trunk_prev_rot[b * 9 + s] = v;
    }
  }
}

// ================================================================================================
// The PACKED ORTH kernel (round 3): FOUR instances per wavefront for the task problems whose only constraints are the stance feet's
// contact equalities — BASELINE configs[1] (five EE tasks + CoM task + posture, 12 contact rows, no bounds, no inequalities).
// The contact rows are eliminated through the orthonormal null-space basis of contact_presolve_orth (Z = [I; G] S,
// S = L^-T, L L' = I + G'G, G = -K^-1 B; [qd_base; qd_legs] = Z y), which leaves an UNCONSTRAINED problem in n' = 6 + (free DoF outside base
// and stance legs) = 14 / 13 unknowns:  H' = (A Z)'(A Z) + d^2 I,  g' = -(A Z)'b (+ Z'g_posture),  H' y = -g',  qd = Z y.
// lane = 16 r + s: instance r of the wave; s = FK slot / DoF column s and 16 + s / reduced variable s. Stages:
//   FK         level-synchronous over DevPlan.q_fk (all joints: the CoM needs every body), sin / cos two per lane;
//   columns    lane s owns the WORLD Jacobian columns of DoF s and 16 + s, and their CoM-Jacobian columns (subtree sums over the contiguous joint
//              range of the DoF's subtree, Robot_Wrapper4.py:670 / Robot_Wrapper2.py:600-603);
//   basis      G on 12 lanes (adjugate), M = I + G'G on 6, its 6 x 6 Cholesky factor and inverse unrolled on every lane, Z to LDS;
//   tasks      one block of six rows at a time (qpA / qpb order, Robot_Wrapper4.py:1271-1294): the block's rows over [base; legs] -> LDS,
//              A Z for the six base-reduced variables on 12 lanes (3 rows each), H' rows and g' accumulated in registers;
//   solve      the packed kernel's two-column Cholesky sweep fused with the forward substitutions (lane s: e_s; lane 15: g'), then
//              y = -L^-T (L^-1 g') as one dot product per lane against the broadcast L^-1 g' — no matrix ever goes back to LDS.
// An instance with a (nearly) rank-deficient leg block is redone by its own wave on the general path (the ORTH variant's QR) in the tail.
// ================================================================================================
constexpr int QLEV = 6;
struct __attribute__((aligned(16))) QInst {
  double X[272];            // oMi [22][12] -> Kb [12][4] @0, Bb [6][4] @48, G [12][6] @72, M [6][6] @144 -> task block Ab [6][18] @0, AZ [6][16] @108
  double W[136];            // sin / cos [22][2] @0, m c [22][4] @44 -> Z [18][6] @0
  double in[64];            // q [27] @0, ee_target [15] @28, prev_ee_target [15] @43, com_target [3] @58, com_target_vel [3] @61
  double pf[16];            // EE frame origins [5][3]
  double ow[16];            // the EE tasks' reference angular velocities [5][3] (zero without orientation references)
  double cl[32], yv[32];    // the configuration's task weights and gains (wt [96], staged at the top: no global load inside the task loop) ->
  double zv[16], xv[16];    //   Cholesky column pair (entries 16..31 zero); g' -> L^-1 g';  y -> base twist * dt
  double gp[32];            // posture part of g by DoF -> qdot by DoF
};
static_assert(sizeof(QInst) * 4 <= 20480, "8 waves per CU");
static_assert(offsetof(QInst, xv) - offsetof(QInst, cl) == 80 * sizeof(double), "wt [96] = cl | yv | zv | xv");
static_assert(offsetof(WbcConfig, joint_w) - offsetof(WbcConfig, ee_W) == 84 * sizeof(double), "ee_W [30] ee_w [5] ee_gain [30] trunk [13] com_W [3] com_gain [3] joint_w");
constexpr int WT_W = 0, WT_w = 30, WT_G = 35, WT_CW = 78, WT_CG = 81;   // offsets inside wt

#ifdef WBC_ABLATE   // timing cuts 201.. (tools/ablate_orthp.py): the kernel returns after stage k with garbage
#define QSTOP(k, val) do { if (A.dbg_stop == 200 + (k)) { if (valid) { A.out.qdot[(size_t)b * NV + s] = (val); if (s == 0) A.out.status[b] = 0; } return; } } while (0)
#else
#define QSTOP(k, val) do { } while (0)
#endif
// INEQ: the variant for the task problems that keep INEQUALITY rows next to the eliminated contact equalities — trunk box (trunkConstraint,
// Robot_Wrapper4.py:707-754), CoM box (CoMConstraint, :669-694), the velocity box of every DoF (:572-637) — and the trunk task (tests/common.py
// "everything"). In the reduced coordinates y (qd = Z y) the velocity bounds of the base and the stance legs are the ROWS of Z (six columns each),
// the trunk and CoM boxes six dense rows (formed with the task blocks' own A Z machinery), the arm's bounds stay simple bounds: <= 24 rows, two per
// lane, and the packed sim3 kernel's dual active-set method on n' <= 12 unknowns. An instance that needs more than 11 active constraints goes to
// the tail with the flagged ones.
// WARM (INEQ only): working sets in and out — the packed sim3 kernel's scheme (seeds through the add step, x / u rebuilt from the factors, restoration).
// In FULL-problem indexing a row of Z is the velocity bound of its DoF (word 0), the trunk / CoM box rows are findConstraints' rows (word 1).
template <bool INEQ, bool WARM = false>
__global__ void __launch_bounds__(64, 2) wbc_tick_orthp_kernel(const KernelArgs A, const DevModel* __restrict__ models,
                                                               const WbcConfig* __restrict__ cfgs, const DevPlan* __restrict__ plans) {
  __shared__ union { QInst Q[4]; Smem G; } SU;
  const int lane = threadIdx.x, r = lane >> 4, s = lane & 15, rbase = lane & 48;
  QInst& I = SU.Q[r];
  const int b_raw = 4 * blockIdx.x + r;
  const bool valid = b_raw < A.B;
  const int b = valid ? b_raw : A.B - 1;
  int mid = 0;
  if (A.in.model_id) { mid = A.in.model_id[b]; mid = mid < 0 ? 0 : (mid >= A.n_models ? A.n_models - 1 : mid); }
  const DevModel& M = models[mid];
  const WbcConfig& cfg = cfgs[mid];
  const DevPlan& P = plans[mid];
  const double dt = A.dt, inv_dt = 1.0 / A.dt;
  const unsigned long long ws_mine = (INEQ && WARM && s < 2 && A.ws_in && valid) ? A.ws_in[2 * (size_t)b + s] : 0ull;   // (lane 0: bounds, lane 1: rows)
  unsigned long long ws_o0 = 0ull, ws_o1 = 0ull;
  // ---- loads: inputs (coalesced per instance), then the per-lane records
  {
    const double* qg = A.in.q + (size_t)b * NQ;
    const double q0 = qg[s], q1 = (16 + s < NQ) ? qg[16 + s] : 0.0;
    const double et = (s < 15 && A.in.ee_target) ? A.in.ee_target[(size_t)b * 15 + s] : 0.0;
    const double ep = (s < 15 && A.in.prev_ee_target) ? A.in.prev_ee_target[(size_t)b * 15 + s] : 0.0;
    double cm = 0.0;
    if (s < 3) cm = A.in.com_target ? A.in.com_target[(size_t)b * 3 + s] : 0.0;
    else if (s < 6) cm = A.in.com_target_vel ? A.in.com_target_vel[(size_t)b * 3 + (s - 3)] : 0.0;
    I.in[s] = q0;
    if (16 + s < 28) I.in[16 + s] = q1;
    if (s < 15) { I.in[28 + s] = et; I.in[43 + s] = ep; }
    if (s < 6) I.in[58 + s] = cm;
    // calcTargetVelEE3's orientation feed-forward (Robot_Wrapper4.py:1125-1133): omega = vee(((R* - R*_prev) / dt) R*^T), one component per lane
    // (EE s / 3, component s % 3), straight from the caller's [B][5][9] references; zero when none are passed
    {
      double om = 0.0;
      if (A.in.ee_ref_rot && s < 15) {
        const int e = s / 3, i = s - 3 * e;
        const double* Rs = A.in.ee_ref_rot + (size_t)b * 45 + 9 * e;
        const double* Rp = A.in.ee_prev_rot + (size_t)b * 45 + 9 * e;
        const int ra = (i == 0) ? 6 : ((i == 1) ? 0 : 3), rb = (i == 0) ? 3 : ((i == 1) ? 6 : 0);   // S[2][1] = D row 2 . R row 1; S[0][2]; S[1][0]
        om = ((Rs[ra] - Rp[ra]) * inv_dt) * Rs[rb] + ((Rs[ra + 1] - Rp[ra + 1]) * inv_dt) * Rs[rb + 1] + ((Rs[ra + 2] - Rp[ra + 2]) * inv_dt) * Rs[rb + 2];
      }
      I.ow[s] = om;
    }
    // the configuration's weights and gains: 85 contiguous doubles of WbcConfig, six per lane, parked in wt (= cl | yv | zv | xv)
    const double* cw = &cfg.ee_W[0][0];
#pragma unroll
    for (int i = 0; i < 6; ++i) I.cl[s + 16 * i] = (s + 16 * i < 89) ? cw[s + 16 * i] : 0.0;     // (+ trunk_box_z_frac, _ang, _scale, com_box_scale @85..88)
    if (INEQ) {
      const bool c_tr = cfg.task_trunk != 0;
      double t0 = 0.0, t1 = 0.0;
      if (c_tr) {
        auto tinv = [&](const int k) -> double {
          return (k < 3) ? A.in.trunk_target[(size_t)b * 3 + k] : (k < 6) ? A.in.prev_trunk_target[(size_t)b * 3 + (k - 3)]
               : (k < 9) ? A.in.trunk_ref_euler[(size_t)b * 3 + (k - 6)] : A.in.trunk_prev_rot[(size_t)b * 9 + (k - 9)];
        };
        t0 = tinv(s); t1 = (s < 2) ? tinv(16 + s) : 0.0;
      }
      I.gp[s] = t0;
      if (s < 2) I.gp[16 + s] = t1;
      if (s >= 2 && s < 6) I.gp[16 + s] = (cfg.con_trunk && A.in.trunk_box_center) ? A.in.trunk_box_center[(size_t)b * 4 + (s - 2)] : 0.0;
    }
  }
  const double* const wt = I.cl;
  const int nv = M.nv, nq = M.nq, nj = M.njoints, n = P.q_nred, nelim = P.nelim, nl = 3 * nelim;
  int efoot[5];
#pragma unroll
  for (int e = 0; e < 5; ++e) efoot[e] = P.q_efoot[e];
  const DevPlan::QDof D0 = P.q_dof[s], D1 = P.q_dof[16 + s];
  const DevPlan::QJnt Jm0 = P.q_jm[s], Jm1 = P.q_jm[16 + s];
  DevPlan::PkJoint fkn = P.q_fk[0][s];
  const int scq0 = P.q_scq[(2 + s) & 31], scq1 = P.q_scq[(18 + s) & 31];
  const bool has1 = 16 + s < nv;                                  // this lane's second DoF exists
  const bool c_com = cfg.task_com != 0;
  const int c_task_joint = cfg.task_joint;
  const int fjoint = (s < 5) ? M.frame_joint[WBC_FR_EE0 + s] : 1;
  const double fp0 = (s < 5) ? M.frame_p[WBC_FR_EE0 + s][0] : 0.0, fp1 = (s < 5) ? M.frame_p[WBC_FR_EE0 + s][1] : 0.0,
               fp2 = (s < 5) ? M.frame_p[WBC_FR_EE0 + s][2] : 0.0;
  WSYNC();
  const double* const qv = I.in;
  const bool c_trunk = INEQ && cfg.task_trunk != 0, c_con_trunk = INEQ && cfg.con_trunk != 0, c_con_com = INEQ && cfg.con_com != 0;
  if (INEQ && __ballot(c_trunk)) {
    // calcTargetVelTrunk2 (Robot_Wrapper4.py:948-1015) / TrunkB (:914-920), as in the packed sim3 / box kernels: the trunk frame is the free-flyer's own
    // placement (the plan checks it); the target velocity x trunk_w is parked in gp [24..29]
    const double* tw = wt + 65;              // trunk_W [0..5], trunk_w [6], trunk_gain [7..12]
    const double* tin = I.gp;
    const double* xt = tin;
    const double* xp = tin + 3;
    const double* er = tin + 6;
    double* const sh = I.X;                  // (free until the FK)
    double Rt_[9], fq[4], rq[4], Rs[9], vel[6];
    quat_to_R(qv + 3, Rt_);
    R_to_quat(Rt_, fq);
    {
      const SinCos t = sincos_cw(s < 3 ? er[s < 3 ? s : 0] : 0.5 * er[(s < 6 ? s : 3) - 3]);
      if (s < 6) { sh[2 * s] = t.s; sh[2 * s + 1] = t.c; }
      WSYNC();
      const double sa = sh[0], ca = sh[1], sb = sh[2], cb = sh[3], sc_ = sh[4], cc_ = sh[5];
      Rs[0] = cc_ * cb; Rs[1] = cc_ * sb * sa - sc_ * ca; Rs[2] = cc_ * sb * ca + sc_ * sa;
      Rs[3] = sc_ * cb; Rs[4] = sc_ * sb * sa + cc_ * ca; Rs[5] = sc_ * sb * ca - cc_ * sa;
      Rs[6] = -sb;      Rs[7] = cb * sa;                  Rs[8] = cb * ca;
      const double qx[4] = {sh[6], 0, 0, sh[7]}, qy[4] = {0, sh[8], 0, sh[9]}, qz[4] = {0, 0, sh[10], sh[11]};
      double tq[4];
      quat_mul(qy, qx, tq);
      quat_mul(qz, tq, rq);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) vel[i] = (xt[i] - xp[i]) * inv_dt + tw[7 + i] * ((xt[i] - qv[i]) * inv_dt);
    const double qe0 = fq[3] * rq[0] - fq[0] * rq[3] + fq[1] * rq[2] - fq[2] * rq[1];   // :974
    const double qe1 = fq[3] * rq[1] - fq[1] * rq[3] - fq[0] * rq[2] + fq[2] * rq[0];   // :975
    const double qe2 = fq[3] * rq[2] - fq[3] * rq[2] + fq[0] * rq[1] - fq[1] * rq[0];   // :976 (sic)
    double D[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) D[i] = (Rs[i] - tin[9 + i]) * inv_dt;
    vel[3] = (D[6] * Rs[1] + D[7] * Rs[4] + D[8] * Rs[7]) + tw[10] * qe0;
    vel[4] = (D[0] * Rs[2] + D[1] * Rs[5] + D[2] * Rs[8]) + tw[11] * qe1;
    vel[5] = (D[3] * Rs[0] + D[4] * Rs[3] + D[5] * Rs[6]) + tw[12] * qe2;
    const double trunk_w = tw[6];
    if (s == 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i) I.gp[24 + i] = c_trunk ? vel[i] * trunk_w : 0.0;
    }
    WSYNC();
  }
  double* const oMi = I.X;                   // [22][12]
  double* const sc = I.W;                    // sin / cos of joint j at 2 j
  double* const mc = I.W + 44;               // m_j c_j (world), m_j at 4 j
  {
    if (scq0 >= 0) { const SinCos t = sincos_cw(qv[scq0]); sc[2 * (2 + s)] = t.s; sc[2 * (2 + s) + 1] = t.c; }
    if (scq1 >= 0) { const SinCos t = sincos_cw(qv[scq1]); sc[2 * (18 + s)] = t.s; sc[2 * (18 + s) + 1] = t.c; }
    if (s == 0) {   // root free-flyer (joint 1): R from the quaternion as Eigen's toRotationMatrix, p = xyz; R column-major then p
      double Rt[9];
      quat_to_R(qv + 3, Rt);
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) oMi[12 + 3 * c + rr] = Rt[3 * rr + c];
      oMi[12 + 9] = qv[0]; oMi[12 + 10] = qv[1]; oMi[12 + 11] = qv[2];
    }
  }
  WSYNC();
  QSTOP(1, sc[4 + s] + oMi[12 + s]);
  // ---- pin.forwardKinematics, level by level (Robot_Wrapper4.py:400)
#pragma unroll 1
  for (int L = 0; L < QLEV; ++L) {
    const DevPlan::PkJoint fk = fkn;
    if (L + 1 < QLEV) fkn = P.q_fk[L + 1][s];
    const int j = fk.joint;
    if (j >= 0) {
      const bool rev = fk.rev != 0;
      const int a0 = fk.a0, a1 = fk.a1, a2 = fk.a2;
      const double* Pp = oMi + 12 * fk.parent;
      const double sn = rev ? sc[2 * j] : 0.0, cs = rev ? sc[2 * j + 1] : 1.0;
      const double pris = rev ? 0.0 : qv[fk.q_idx];
      double Av[3], Bv[3], Cv[3], Pv[3];
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) { Av[rr] = Pp[a0 + rr]; Bv[rr] = Pp[a1 + rr]; Cv[rr] = Pp[a2 + rr]; Pv[rr] = Pp[9 + rr]; }
      double* Po = oMi + 12 * j;
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        Po[a0 + rr] = Av[rr];
        Po[a1 + rr] = cs * Bv[rr] + sn * Cv[rr];
        Po[a2 + rr] = cs * Cv[rr] - sn * Bv[rr];
        Po[9 + rr] = Pv[rr] + Av[rr] * (fk.t0 + pris) + Bv[rr] * fk.t1 + Cv[rr] * fk.t2;
      }
    }
    WSYNC();
  }
  QSTOP(2, oMi[12 * 4 + (s & 7)]);
  // ---- frame origins (updateFramePlacements, :405), m c per joint, Jacobian columns (WORLD) of DoF s and 16 + s
  double ms_l = 0.0, sl[3] = {0, 0, 0};
  {
    if (s < 5) {
      const double* Pg = oMi + 12 * fjoint;
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) I.pf[3 * s + rr] = Pg[9 + rr] + Pg[rr] * fp0 + Pg[3 + rr] * fp1 + Pg[6 + rr] * fp2;
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int j = s + 16 * hh;
      const DevPlan::QJnt& Jm = hh ? Jm1 : Jm0;
      if (j >= 1 && j < nj) {
        const double* Pj = oMi + 12 * j;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          const double v = Jm.m * (Pj[9 + rr] + Pj[rr] * Jm.c0 + Pj[3 + rr] * Jm.c1 + Pj[6 + rr] * Jm.c2);
          mc[4 * j + rr] = v; sl[rr] += v;
        }
        mc[4 * j + 3] = Jm.m; ms_l += Jm.m;
      }
    }
  }
  double lin0[3] = {0, 0, 0}, ang0[3] = {0, 0, 0}, lin1[3] = {0, 0, 0}, ang1[3] = {0, 0, 0};
  {
    const double* Pj = oMi + 12 * D0.joint;
    const double pj[3] = {Pj[9], Pj[10], Pj[11]};
    if (D0.ang >= 0) { ang0[0] = Pj[3 * D0.ang]; ang0[1] = Pj[3 * D0.ang + 1]; ang0[2] = Pj[3 * D0.ang + 2]; cross3(pj, ang0, lin0); }
    if (D0.lin >= 0) { lin0[0] = Pj[3 * D0.lin]; lin0[1] = Pj[3 * D0.lin + 1]; lin0[2] = Pj[3 * D0.lin + 2]; }
  }
  if (has1) {
    const double* Pj = oMi + 12 * D1.joint;
    const double pj[3] = {Pj[9], Pj[10], Pj[11]};
    if (D1.ang >= 0) { ang1[0] = Pj[3 * D1.ang]; ang1[1] = Pj[3 * D1.ang + 1]; ang1[2] = Pj[3 * D1.ang + 2]; cross3(pj, ang1, lin1); }
    if (D1.lin >= 0) { lin1[0] = Pj[3 * D1.lin]; lin1[1] = Pj[3 * D1.lin + 1]; lin1[2] = Pj[3 * D1.lin + 2]; }
  }
  WSYNC();   // oMi is dead: X is free
  // ---- centre of mass and the CoM-Jacobian columns (pin.jacobianCenterOfMass): jc = (m_sub / M) (lin + ang x c_sub)
  double com[3] = {0, 0, 0}, jc0[3] = {0, 0, 0}, jc1[3] = {0, 0, 0};
  if (__ballot(c_com || c_con_com)) {
    const double Mt = rsum16(ms_l);
    const double St[3] = {rsum16(sl[0]), rsum16(sl[1]), rsum16(sl[2])};
    com[0] = St[0] / Mt; com[1] = St[1] / Mt; com[2] = St[2] / Mt;
    auto jcom = [&](const DevPlan::QDof& D, const double* lin, const double* ang, const bool on, double* jc) {
      double ms = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0;
      if (D.joint == 1) { ms = Mt; s0 = St[0]; s1 = St[1]; s2 = St[2]; }          // the free-flyer moves every body
      else {
#pragma unroll
        for (int t = 0; t < 8; ++t) {                                             // (sub-trees of at most 8 joints: checked on the host)
          const int j = D.sub_lo + t;
          if (on && j <= D.sub_hi) {
            const double2a m0 = lds2(mc + 4 * j), m1 = lds2(mc + 4 * j + 2);
            s0 += m0.x; s1 += m0.y; s2 += m1.x; ms += m1.y;
          }
        }
      }
      if (on && ms > 0.0) {
        const double cs_[3] = {s0 / ms, s1 / ms, s2 / ms};
        double wxc[3];
        cross3(ang, cs_, wxc);
        const double f = ms / Mt;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) jc[rr] = f * (lin[rr] + wxc[rr]);
      }
    };
    jcom(D0, lin0, ang0, true, jc0);
    jcom(D1, lin1, ang1, has1, jc1);
  }
  QSTOP(3, lin0[0] + ang0[1] + lin1[2] + jc0[0] + jc1[1] + com[2]);
  // ---- contact rows (EEConstraint, :757-761: WORLD linear rows): K (leg DoF) and B (base DoF) through LDS, G = -K^-1 B
  double* const Kb = I.X;                    // [12][4]
  double* const Bb = I.X + 48;               // [6][4]
  double* const Gm = I.X + 72;               // [12][6]
  double* const Mm = I.X + 144;              // [6][6]
  {
    if (D0.bl >= 6) { Kb[4 * (D0.bl - 6)] = lin0[0]; Kb[4 * (D0.bl - 6) + 1] = lin0[1]; Kb[4 * (D0.bl - 6) + 2] = lin0[2]; }
    else if (D0.bl >= 0) { Bb[4 * D0.bl] = lin0[0]; Bb[4 * D0.bl + 1] = lin0[1]; Bb[4 * D0.bl + 2] = lin0[2]; }
    if (has1 && D1.bl >= 6) { Kb[4 * (D1.bl - 6)] = lin1[0]; Kb[4 * (D1.bl - 6) + 1] = lin1[1]; Kb[4 * (D1.bl - 6) + 2] = lin1[2]; }
  }
  WSYNC();
  bool defer = false;
  {
    double grow[6] = {0, 0, 0, 0, 0, 0};
    const int f = (s < nl) ? s / 3 : 0, i = (s < nl) ? s - 3 * f : 0;
    const double* k0 = Kb + 4 * (3 * f); const double* k1 = k0 + 4; const double* k2 = k1 + 4;
    const double k00 = k0[0], k10 = k0[1], k20 = k0[2], k01 = k1[0], k11 = k1[1], k21 = k1[2], k02 = k2[0], k12 = k2[1], k22 = k2[2];
    const double a00 = k11 * k22 - k12 * k21, a01 = k02 * k21 - k01 * k22, a02 = k01 * k12 - k02 * k11;
    const double a10 = k12 * k20 - k10 * k22, a11 = k00 * k22 - k02 * k20, a12 = k02 * k10 - k00 * k12;
    const double a20 = k10 * k21 - k11 * k20, a21 = k01 * k20 - k00 * k21, a22 = k00 * k11 - k01 * k10;
    const double det = k00 * a00 + k01 * a10 + k02 * a20;
    const double sc_ = fabs(k00) + fabs(k01) + fabs(k02) + fabs(k10) + fabs(k11) + fabs(k12) + fabs(k20) + fabs(k21) + fabs(k22);
    const bool bad = (s < nl) && !(fabs(det) > fmax(1e-6, A.sing_tol) * sc_ * sc_ * sc_);      // (orth_null_basis' bar: below it the QR decides)
    defer = ((__ballot(bad) >> rbase) & 0xFFFFull) != 0 || (A.orth_qr != 0);
    const double id = -1.0 / det;
    const double r0 = (i == 0) ? a00 : (i == 1) ? a10 : a20, r1 = (i == 0) ? a01 : (i == 1) ? a11 : a21, r2 = (i == 0) ? a02 : (i == 1) ? a12 : a22;
    if (s < nl) {
#pragma unroll
      for (int c = 0; c < 6; ++c) grow[c] = id * (r0 * Bb[4 * c] + r1 * Bb[4 * c + 1] + r2 * Bb[4 * c + 2]);
    }
    if (s < 12) {
#pragma unroll
      for (int c = 0; c < 6; c += 2) sts2(Gm + s * 6 + c, grow[c], grow[c + 1]);   // (rows >= nl: zero)
    }
  }
  WSYNC();
  // ---- M = I + G'G (lane s < 6: row s) and its Cholesky factor L L' = M, COOPERATIVELY: lane r keeps row r of L in six registers, a finished
  // row and the reciprocal of its pivot go through LDS (M >= I: no pivot can fail). Unrolled on every lane the factor and its inverse took
  // 66+ VGPRs and pushed ~20 live values out to scratch — 0.35 GB of spill traffic per 65536-tick launch (FETCH_SIZE / WRITE_SIZE).
  double* const Zm = I.W;                    // [18][6]: rows 0..5 base DoF, 6 + l eliminated leg DoF l
  double xc[6];
  {
    double Lr[6] = {0, 0, 0, 0, 0, 0};
    if (s < 6) {
#pragma unroll
      for (int l = 0; l < 12; ++l) {
        const double gl = Gm[l * 6 + s];
        const double2a t0 = lds2(Gm + l * 6), t1 = lds2(Gm + l * 6 + 2), t2 = lds2(Gm + l * 6 + 4);
        Lr[0] = fma(gl, t0.x, Lr[0]); Lr[1] = fma(gl, t0.y, Lr[1]); Lr[2] = fma(gl, t1.x, Lr[2]);
        Lr[3] = fma(gl, t1.y, Lr[3]); Lr[4] = fma(gl, t2.x, Lr[4]); Lr[5] = fma(gl, t2.y, Lr[5]);
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) Lr[k] += (k == s) ? 1.0 : 0.0;
    }
    double* const Lq = Mm;                   // finished rows of L [6][6], then 1 / L_jj at [36 + j]
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double v = Lr[j];
#pragma unroll
      for (int k = 0; k < j; ++k) v = fma(-Lr[k], Lr[k], v);
      v = (s == j) ? v : 1.0;
      double rs = __builtin_amdgcn_rsq(v);
      rs = rs * fma(-0.5 * v * rs, rs, 1.5); rs = rs * fma(-0.5 * v * rs, rs, 1.5);
      if (s == j) {
        Lr[j] = v * rs;
#pragma unroll
        for (int k = 0; k <= j; ++k) Lq[j * 6 + k] = Lr[k];
        Lq[36 + j] = rs;
      }
      WSYNC();
      if (j < 5) {
        double w = Lr[j];
#pragma unroll
        for (int k = 0; k < j; ++k) w = fma(-Lr[k], Lq[j * 6 + k], w);
        if (s > j && s < 6) Lr[j] = w * Lq[36 + j];
      }
    }
    // lane c = min(s, 5): column c of L^-1 (L x = e_c; entries above c are zero) = row c of S = L^-T, the base part of Z
    const int cs = s < 6 ? s : 5;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      double w = (i == cs) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) w = fma(-Lq[i * 6 + k], xc[k], w);
      xc[i] = (i < cs) ? 0.0 : w * Lq[36 + i];
    }
    WSYNC();                                 // (the m c table in W has been read by everyone: W becomes Z)
    if (s < 6) { sts2(Zm + s * 6, xc[0], xc[1]); sts2(Zm + s * 6 + 2, xc[2], xc[3]); sts2(Zm + s * 6 + 4, xc[4], xc[5]); }   // S[s][k] = Li[k][s]
    WSYNC();
    if (s < 12) {                            // Z_leg row s = G row s times S
      const double2a a = lds2(Gm + s * 6), bq = lds2(Gm + s * 6 + 2), c = lds2(Gm + s * 6 + 4);
      const double grow[6] = {a.x, a.y, bq.x, bq.y, c.x, c.y};
      double o[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int c2 = 0; c2 < 6; ++c2) {
        const double2a z0 = lds2(Zm + c2 * 6), z1 = lds2(Zm + c2 * 6 + 2), z2 = lds2(Zm + c2 * 6 + 4);
        o[0] = fma(grow[c2], z0.x, o[0]); o[1] = fma(grow[c2], z0.y, o[1]); o[2] = fma(grow[c2], z1.x, o[2]);
        o[3] = fma(grow[c2], z1.y, o[3]); o[4] = fma(grow[c2], z2.x, o[4]); o[5] = fma(grow[c2], z2.y, o[5]);
      }
      double* zr = Zm + (6 + s) * 6;
      sts2(zr, o[0], o[1]); sts2(zr + 2, o[2], o[3]); sts2(zr + 4, o[4], o[5]);
    }
  }
  WSYNC();
  QSTOP(4, Zm[s * 6 + 1] + lin0[0] + lin1[1] + jc0[2] + jc1[0]);
  // ---- the task stack, one block of six rows at a time
  const int hh = s >> 3, cc = s & 7;         // A Z: lane (hh, cc < 6) forms rows 3 hh .. 3 hh + 2 of the block for base-reduced variable cc
  // (column cc of Z is re-read from LDS inside each block: kept in 36 registers across the task loop it pushed the row of H' out to scratch —
  //  33 spill instructions per wave, 1.3 % of the step)
  const double* const Zcol = Zm + (cc < 6 ? cc : 0);
#define ZCJ(j) Zcol[(j) * 6]
  double h[16], gacc = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) h[k] = 0.0;
  double* const Ab = I.X;                    // [6][18]
  double* const AZ = I.X + 108;              // [6][16]
  auto block = [&](const double* a0, const double* a1, const double* br, const int ef, const bool dense, const bool arm, const bool acc_h = true) {
    WSYNC();                                 // the previous block's readers are done
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) {
      if (D0.bl >= 0) Ab[rr * 18 + D0.bl] = a0[rr]; else if (D0.red >= 6) AZ[rr * 16 + D0.red] = a0[rr];
      if (has1) { if (D1.bl >= 0) Ab[rr * 18 + D1.bl] = a1[rr]; else if (D1.red >= 6) AZ[rr * 16 + D1.red] = a1[rr]; }
      if (s >= n) AZ[rr * 16 + s] = 0.0;     // padding variables
    }
    WSYNC();
    if (cc < 6) {
      // (the leg window of Z column cc straight from LDS: picked out of the register copy by `ef`, the compiler turned Zc into a scratch array)
      const int jl = 6 + 3 * (ef < 0 ? 0 : ef);
      double zw[3];
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) zw[jj] = (ef >= 0) ? Zm[(jl + jj) * 6 + cc] : 0.0;
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        const double* row = Ab + (3 * hh + rr) * 18;
        double acc = 0.0;
        if (dense) {
#pragma unroll
          for (int j = 0; j < 18; j += 2) { const double2a v = lds2(row + j); acc = fma(v.x, ZCJ(j), fma(v.y, ZCJ(j + 1), acc)); }
        } else {
#pragma unroll
          for (int j = 0; j < 6; j += 2) { const double2a v = lds2(row + j); acc = fma(v.x, ZCJ(j), fma(v.y, ZCJ(j + 1), acc)); }
          acc = fma(row[jl], zw[0], fma(row[jl + 1], zw[1], fma(row[jl + 2], zw[2], acc)));
        }
        AZ[(3 * hh + rr) * 16 + cc] = acc;
      }
    }
    WSYNC();
    if (!acc_h) return;                      // (INEQ: the constraint rows' images stay in AZ)
    double own[6];
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) own[rr] = AZ[rr * 16 + s];
    gacc = fma(-own[0], br[0], fma(-own[1], br[1], fma(-own[2], br[2], fma(-own[3], br[3], fma(-own[4], br[4], fma(-own[5], br[5], gacc))))));
    if (arm) {
#pragma unroll
      for (int k = 0; k < 16; k += 2) {
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) { const double2a v = lds2(AZ + rr * 16 + k); h[k] = fma(own[rr], v.x, h[k]); h[k + 1] = fma(own[rr], v.y, h[k + 1]); }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 6; k += 2) {
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) { const double2a v = lds2(AZ + rr * 16 + k); h[k] = fma(own[rr], v.x, h[k]); h[k + 1] = fma(own[rr], v.y, h[k + 1]); }
      }
    }
  };
  // (the CoM block goes first — H' is a sum, the order is free — so that its Jacobian columns are dead before the EE loop: live across it they
  //  were spilled around the loop, 20 values per lane per wave)
  if (__ballot(c_com)) {   // Robot_Wrapper2 comJacobian (:600-603), cartesianTargetCoM (:661-668)
    double a0[6] = {0, 0, 0, 0, 0, 0}, a1[6] = {0, 0, 0, 0, 0, 0}, br[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
      const double cw = wt[WT_CW + rr];
      a0[rr] = cw * jc0[rr]; a1[rr] = has1 ? cw * jc1[rr] : 0.0;
      br[rr] = I.in[61 + rr] + wt[WT_CG + rr] * (I.in[58 + rr] - com[rr]);
    }
    block(a0, a1, br, -1, true, true);
  }
  const unsigned tmask = (unsigned)__builtin_amdgcn_readfirstlane((int)P.task_ee_mask);
  const unsigned armsup = (unsigned)__builtin_amdgcn_readfirstlane((int)P.q_armsup);
#pragma unroll 1
  for (unsigned tm = tmask; tm; tm &= tm - 1) {   // endEffectorA2 (:474-484) / calcTargetVelEE3 (:1052-1157) / EndEffectorB2 (:907-910)
    const int e = __ffs((int)tm) - 1;
    const double w = wt[WT_w + e];
    double Wd[6], Gd[3];
#pragma unroll
    for (int i = 0; i < 6; ++i) Wd[i] = wt[WT_W + 6 * e + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) Gd[i] = wt[WT_G + 6 * e + i];
    const double pfe[3] = {I.pf[3 * e], I.pf[3 * e + 1], I.pf[3 * e + 2]};
    const bool sup0 = (D0.supmask >> e) & 1, sup1 = has1 && ((D1.supmask >> e) & 1);
    double a0[6], a1[6], br[6] = {0, 0, 0, 0, 0, 0}, wxp[3];
    cross3(ang0, pfe, wxp);
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) { a0[rr] = sup0 ? Wd[rr] * ((lin0[rr] + wxp[rr]) * w) : 0.0; a0[3 + rr] = sup0 ? Wd[3 + rr] * (ang0[rr] * w) : 0.0; }
    cross3(ang1, pfe, wxp);
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) { a1[rr] = sup1 ? Wd[rr] * ((lin1[rr] + wxp[rr]) * w) : 0.0; a1[3 + rr] = sup1 ? Wd[3 + rr] * (ang1[rr] * w) : 0.0; }
    const double* xt = I.in + 28 + 3 * e;
    const double* xp = I.in + 43 + 3 * e;
#pragma unroll
    for (int i = 0; i < 3; ++i) { br[i] = ((xt[i] - xp[i]) * inv_dt + Gd[i] * ((xt[i] - pfe[i]) * inv_dt)) * w; br[3 + i] = I.ow[3 * e + i] * w; }
    int ef = efoot[0];
#pragma unroll
    for (int i = 1; i < 5; ++i) ef = (e == i) ? efoot[i] : ef;
    block(a0, a1, br, ef, false, (armsup >> e) & 1u);
  }
  // ---- INEQ: the trunk task (trunkA, Robot_Wrapper4.py:487-490, WORLD rows on the base columns), then the inequality rows and every bound
  double lb = -QP_INF, ub = QP_INF, clb0 = -QP_INF, cub0 = QP_INF, clb1 = -QP_INF, cub1 = QP_INF;
  double rowreg[6] = {0, 0, 0, 0, 0, 0};
  if (INEQ) {
    if (__ballot(c_trunk)) {
      const double* tw = wt + 65;
      const double trunk_w = tw[6];
      const bool sup = c_trunk && D0.bl >= 0 && D0.bl < 6;
      double at[6], a1[6] = {0, 0, 0, 0, 0, 0}, br[6];
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        at[rr] = sup ? (tw[rr] * lin0[rr]) * trunk_w : 0.0;
        at[3 + rr] = sup ? (tw[3 + rr] * ang0[rr]) * trunk_w : 0.0;
      }
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) br[rr] = c_trunk ? I.gp[24 + rr] : 0.0;
      block(at, a1, br, -1, false, false);
    }
    // the six dense rows in reduced coordinates, through the blocks' A Z machinery: rows 0..3 trunk box (z, roll, pitch, yaw: LOCAL_WORLD_ALIGNED rows
    // of the trunk frame = the free-flyer's placement, :707-754), rows 4, 5 CoM box (x, y rows of the CoM Jacobian, :669-694)
    const double ptr[3] = {qv[0], qv[1], qv[2]};
    {
      double a0[6] = {0, 0, 0, 0, 0, 0}, a1[6] = {0, 0, 0, 0, 0, 0}, zr[6] = {0, 0, 0, 0, 0, 0};
      if (c_con_trunk && D0.bl >= 0 && D0.bl < 6) {
        double wxp[3];
        cross3(ang0, ptr, wxp);
        a0[0] = lin0[2] + wxp[2]; a0[1] = ang0[0]; a0[2] = ang0[1]; a0[3] = ang0[2];
      }
      if (c_con_com) { a0[4] = jc0[0]; a0[5] = jc0[1]; a1[4] = has1 ? jc1[0] : 0.0; a1[5] = has1 ? jc1[1] : 0.0; }
      block(a0, a1, zr, -1, true, true, false);          // (the images stay in AZ through the sweep, which leaves X alone)
    }
    // bounds: the velocity damper of this lane's two DoF (:572-637) goes to its row of Z (base / stance leg) or to its reduced variable (arm); the
    // trunk box's and the CoM box's sides to rows 0..5
    double* const rbl = I.X + 204;            // row bounds [32] lower, [32] upper
    double* const rbu = I.X + 236;
    double* const vbl = I.pf;                 // simple bounds of the reduced variables [16] lower / upper (pf | ow are dead)
    double* const vbu = I.ow;
    double l0 = 0.0, u0 = 0.0, l1 = 0.0, u1 = 0.0, tl = -QP_INF, tu = QP_INF;
    {
      const double dcoef = cfg.damper_coef, dqi = cfg.damper_qi, dqs = cfg.damper_qs;
      auto damper = [&](const DevPlan::XVar& v, double& l_, double& u_) {
        const double qi = qv[v.dq_idx], lo = v.d_lo, hi = v.d_hi, vm = v.d_vm;
        if (qi <= lo + dqi) { l_ = -dcoef * (qi - lo - dqs) / (dqi - dqs); if (l_ > vm) l_ = vm; if (l_ < -vm) l_ = -vm; } else l_ = -vm;
        if (qi >= hi - dqi) { u_ = dcoef * (hi - qi - dqs) / (dqi - dqs); if (u_ < -vm) u_ = -vm; if (u_ > vm) u_ = vm; } else u_ = vm;
        if (l_ > 0) l_ = -l_;
        if (u_ < 0) u_ = -u_;
      };
      const DevPlan::XVar v0 = P.q_dmp[s], v1 = P.q_dmp[(16 + s) & 31];
      damper(v0, l0, u0);
      if (has1) damper(v1, l1, u1);
      // trunk box (lanes 0..3) and CoM box (lanes 4, 5)
      const double tb_z = wt[85], tb_a = wt[86], tb_s = wt[87], cb_s = wt[88];
      double Rtr[9];
      quat_to_R(qv + 3, Rtr);
      const double ay = (s == 1) ? Rtr[7] : ((s == 2) ? -Rtr[6] : Rtr[3]);
      const double ax = (s == 1) ? Rtr[8] : ((s == 2) ? sqrt(fma(Rtr[7], Rtr[7], Rtr[8] * Rtr[8])) : Rtr[0]);
      const double eul = atan2(ay, ax);       // lanes 1, 2, 3 hold roll, pitch, yaw
      const double* bc = I.gp + 18;
      if (c_con_trunk && s < 4) {
        const double cr = (s == 0) ? ptr[2] : eul;
        const double vr = (s == 0) ? bc[0] * tb_z : tb_a;
        tl = (((bc[s] - vr) - cr) * inv_dt) * tb_s;
        tu = (((bc[s] + vr) - cr) * inv_dt) * tb_s;
      }
      if (c_con_com && (s == 4 || s == 5)) {  // EE_frame_pos[1] = FL, [2] = RR (:675-677)
        const int r_ = s - 4;
        tl = ((I.pf[3 * 2 + r_] - com[r_]) * inv_dt) * cb_s;
        tu = ((I.pf[3 * 1 + r_] - com[r_]) * inv_dt) * cb_s;
      }
    }
    WSYNC();                                 // (pf has been read)
    rbl[s] = (s < 6) ? tl : -QP_INF; rbu[s] = (s < 6) ? tu : QP_INF; rbl[16 + s] = -QP_INF; rbu[16 + s] = QP_INF;
    vbl[s] = -QP_INF; vbu[s] = QP_INF;
    WSYNC();
    if (D0.bl >= 0) { rbl[6 + D0.bl] = l0; rbu[6 + D0.bl] = u0; } else if (D0.red >= 6) { vbl[D0.red & 15] = l0; vbu[D0.red & 15] = u0; }
    if (has1) { if (D1.bl >= 0) { rbl[6 + D1.bl] = l1; rbu[6 + D1.bl] = u1; } else if (D1.red >= 6) { vbl[D1.red & 15] = l1; vbu[D1.red & 15] = u1; } }
    // (rows and bounds wait in LDS — X beyond AZ, pf | ow — until the QP: read into registers before the sweep they were spilled across it,
    //  ~1.3 KB of scratch traffic per tick)
  }
  QSTOP(5, h[0] + h[5] + h[13] + gacc);
  // posture rows (qpJointA / qpJointb, :1199-1268): Z'(d^2 I)Z = d^2 I; the target's part of g through Z
  const double joint_w = wt[84];
  const double dpost = (1.0 / nv) * joint_w;
  if (__ballot(c_task_joint == WBC_JOINT_PREV)) {
    WSYNC();
    const bool prev = c_task_joint == WBC_JOINT_PREV;
    I.gp[s] = prev ? -dpost * ((1.0 / nv) * qv[s < 6 ? s : s + 1] * joint_w) : 0.0;
    I.gp[16 + s] = (prev && has1) ? -dpost * ((1.0 / nv) * qv[17 + s] * joint_w) : 0.0;
    WSYNC();
    if (s < 6) {
#pragma unroll
      for (int j = 0; j < 18; ++j) gacc = fma(ZCJ(j), I.gp[P.q_bl2dof[j] & 31], gacc);
    } else if (s < n) gacc += I.gp[P.q_red2dof[s] & 31];
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) if (k == s) h[k] += (s < n) ? dpost * dpost : 1.0;
  if (s >= n) gacc = 0.0;
  bool live = valid && !defer;
  // ---- Cholesky H' = L L' fused with the substitutions: lane s: L y = e_s (row s of L^-T); lane 15 (a padding variable): L y = g'
  WSYNC();                                   // (wt is dead: its memory becomes the sweep's vectors, zero beyond entry 15)
  I.cl[16 + s] = 0.0; I.yv[16 + s] = 0.0;
  I.zv[s] = gacc;
  WSYNC();
  double y[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) y[k] = (s == 15) ? I.zv[k] : ((k == s) ? 1.0 : 0.0);
  double pmin = 1.0;
#pragma unroll 1
  for (int j = 0; j < 16; j += 2) {
    WSYNC();
    I.cl[s] = h[0]; I.yv[s] = h[1];
    WSYNC();
    const double* c0 = I.cl + j;
    const double* c1 = I.yv + j;
    double cm0[16], cm1[16];
#pragma unroll
    for (int rr = 0; rr < 16; rr += 2) {
      const double2a v0 = lds2(c0 + rr), v1 = lds2(c1 + rr);
      cm0[rr] = v0.x; cm0[rr + 1] = v0.y; cm1[rr] = v1.x; cm1[rr + 1] = v1.y;
    }
    const double pj = cm0[0];
    pmin = (pj > 0.0) ? fmin(pmin, pj) : -1.0;
    const double rinv = rsqrt(pj), ipj = rinv * rinv;
    const double th = h[0] * ipj, ty = y[0] * ipj, yk = y[0] * rinv;
    const double h1 = fma(-th, cm0[1], h[1]), y1 = fma(-ty, cm0[1], y[1]);
    const double a = cm0[1];
#pragma unroll
    for (int rr = 1; rr < 16; ++rr) cm1[rr] = fma(-(cm0[rr] * ipj), a, cm1[rr]);
    const double pj2 = cm1[1];
    pmin = (pj2 > 0.0) ? fmin(pmin, pj2) : -1.0;
    const double rinv2 = rsqrt(pj2), ipj2 = rinv2 * rinv2;
    const double th2 = h1 * ipj2, ty2 = y1 * ipj2, yk2 = y1 * rinv2;
#pragma unroll
    for (int rr = 2; rr < 16; ++rr) h[rr - 2] = fma(-th2, cm1[rr], fma(-th, cm0[rr], h[rr]));
#pragma unroll
    for (int rr = 2; rr < 16; ++rr) y[rr - 2] = fma(-ty2, cm1[rr], fma(-ty, cm0[rr], y[rr]));
    y[14] = fma(-ty2, 0.0, yk); y[15] = yk2;
    h[14] = 0.0; h[15] = 0.0;
  }
  QSTOP(6, y[0] + y[15] + h[0]);
  int status = WBC_QP_OPTIMAL;
  if (!(pmin > 0.0)) status = WBC_QP_NUMERICAL;
  WSYNC();
  if (s == 15) {
#pragma unroll
    for (int k = 0; k < 16; k += 2) sts2(I.zv + k, y[k], y[k + 1]);
  }
  WSYNC();
  double x = 0.0;
  {
    double xa = 0.0, xb = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k += 2) { const double2a v2 = lds2(I.zv + k); xa = fma(y[k], v2.x, xa); xb = fma(y[k + 1], v2.y, xb); }
    x = (s < n) ? -(xa + xb) : 0.0;
  }
  int iters = 0;
  if (INEQ) {
    // ================================ the QP (INEQ variant): the packed sim3 kernel's dual active-set method =====================================
    // unknowns: the n' <= 12 reduced variables (lane s); simple bounds on the arm's; rows 0..5 dense (Cd [6][16]), rows 6 + j = row j of Z (six
    // base-reduced columns): lane s owns rows s and 16 + s (< 24). Codes: bound of variable i = i, row rr = 32 + rr, upper side + 256.
    constexpr int QPV = 12, QLD = 14, QTC = 11;
    double* const J = I.X;                   // [12][14]
    double* const Cd = I.X + 168;            // [6][16]
    double* const T = I.in;                  // [11][14] (runs through in | pf | ow | cl | yv: all dead)
    double* const qxv = I.zv;
    double* const qdv = I.xv;
    double* const qyv = I.gp;
    double* const qtv = I.gp + 16;
    WSYNC();
    {
      const double* rbl = I.X + 204; const double* rbu = I.X + 236;
      lb = I.pf[s]; ub = I.ow[s]; clb0 = rbl[s]; cub0 = rbu[s]; clb1 = rbl[16 + s]; cub1 = rbu[16 + s];
#pragma unroll
      for (int i = 0; i < 6; ++i) rowreg[i] = I.X[108 + s + 16 * i];
    }
    WSYNC();                                 // (everything the QP's matrices overwrite has been read)
    if (s < QPV) {
#pragma unroll
      for (int k = 0; k < QPV; k += 2) sts2(J + s * QLD + k, y[k], y[k + 1]);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) Cd[s + 16 * i] = rowreg[i];
    if (s < QTC) {
#pragma unroll
      for (int k = 0; k < QLD; k += 2) sts2(T + s * QLD + k, 0.0, 0.0);
    }
    double sq = 0.0;
#pragma unroll
    for (int k = 0; k < QPV; ++k) sq = fma(y[k], y[k], sq);
    const double jf2 = rsum16(s < QPV ? sq : 0.0);
    WSYNC();
    auto row_dot = [&](const int rr, const double* v) -> double {      // row rr (0..23) times a vector of the reduced variables (LDS)
      double a = 0.0;
      if (rr < 6) {
#pragma unroll
        for (int k = 0; k < QPV; k += 2) { const double2a c = lds2(Cd + rr * 16 + k), w = lds2(v + k); a = fma(c.x, w.x, fma(c.y, w.y, a)); }
      } else {
        const double* z = Zm + (rr - 6) * 6;
        const double2a c0 = lds2(z), c1 = lds2(z + 2), c2 = lds2(z + 4), w0 = lds2(v), w1 = lds2(v + 2), w2 = lds2(v + 4);
        a = fma(c0.x, w0.x, fma(c0.y, w0.y, fma(c1.x, w1.x, fma(c1.y, w1.y, fma(c2.x, w2.x, c2.y * w2.y)))));
      }
      return a;
    };
    auto row_n2 = [&](const int rr) -> double {
      double a = 0.0;
      if (rr < 6) {
#pragma unroll
        for (int k = 0; k < QPV; ++k) { const double c = Cd[rr * 16 + k]; a = fma(c, c, a); }
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) { const double c = Zm[(rr - 6) * 6 + k]; a = fma(c, c, a); }
      }
      return a;
    };
    const bool has_b = s < n, has_r1 = s < 8;
    const double cn0 = row_n2(s), cn1 = has_r1 ? row_n2(16 + s) : 1.0;
    if (live && ((has_b && ((lb != lb) || (ub != ub))) || (clb0 != clb0) || (cub0 != cub0) || (has_r1 && ((clb1 != clb1) || (cub1 != cub1))))) status = WBC_QP_NUMERICAL;
    {
      const unsigned long long nb = __ballot(status != WBC_QP_OPTIMAL);
      if ((nb >> rbase) & 0xFFFFull) status = WBC_QP_NUMERICAL;
    }
    int actm = 0;                             // bit 0: this lane's bound is active, bit 1: its row s, bit 2: its row 16 + s (ONE register: three bools
                                              // set through a selected reference lived in scratch, a flat store per working-set change)
    bool overflow = false;
    double u = 0.0;
    int a_code = 0, q = 0;
    const int max_iter = 10 * (n + 24) + 20;
    bool searching = live && status == WBC_QP_OPTIMAL;
    const int sJ = s < QPV ? s : QPV - 1, sT = s < QTC ? s : QTC - 1;
    auto normal_d = [&](const bool is_row, const int rr_, const int ip, const double sgn) -> double {
      double d = 0.0;
      if (is_row) {
        if (rr_ < 6) {
#pragma unroll
          for (int k = 0; k < QPV; ++k) d = fma(J[k * QLD + sJ], Cd[rr_ * 16 + k], d);
        } else {
          const double* z = Zm + (rr_ - 6) * 6;
#pragma unroll
          for (int k = 0; k < 6; ++k) d = fma(J[k * QLD + sJ], z[k], d);
        }
        d *= sgn;
      } else d = sgn * J[(ip & 15) * QLD + sJ];
      return d;
    };
    auto set_act = [&](const int code, const bool val) {
      const int rr = code - 32;
      const int bit = (code >= 32) ? ((rr < 16) ? 2 : 4) : 1;
      const bool mine = (code >= 32) ? (s == (rr & 15)) : (s == code);
      if (mine) actm = val ? (actm | bit) : (actm & ~bit);
    };
    auto drop_slot = [&](const bool dr, const int l_) {
      const int l = dr ? l_ : 0;
      const int lc = bpermi(a_code, rbase + l) & 255;
      if (dr) set_act(lc, false);
      WSYNC();
      qyv[s] = u; qtv[s] = (double)a_code;
      WSYNC();
      if (dr && s >= l && s < q - 1) { u = qyv[s + 1]; a_code = (int)qtv[s + 1]; }
      if (dr && s == q - 1) { u = 0.0; a_code = 0; }
      const int srow = (sT >= l) ? ((sT + 1 < QTC) ? sT + 1 : sT) : sT;
      double tx = T[srow * QLD + l];
      double jx = J[sJ * QLD + l];
      double hrun = T[l * QLD + l];
      const int kend = dr ? q - 1 : 0;
#pragma unroll 1
      for (int k0 = 0; k0 < QTC - 1; ++k0) {
        const bool on = dr && (l + k0 < kend);
        if (!__ballot(on)) break;
        const int kk = on ? l + k0 : 0;
        const double tb = T[l * QLD + kk + 1];
        const double nrm2 = fma(hrun, hrun, tb * tb);
        double c_ = 1.0, s_ = 0.0, rho = 0.0;
        if (nrm2 > 0.0) { const double ri = rsqrt(nrm2); c_ = tb * ri; s_ = -hrun * ri; rho = nrm2 * ri; }
        const double ty_ = T[srow * QLD + kk + 1];
        const double jy = J[sJ * QLD + kk + 1];
        WSYNC();
        if (on) {
          hrun = rho;
          if (s < q - 1) T[s * QLD + kk] = fma(c_, tx, s_ * ty_);
          if (has_b) J[s * QLD + kk] = fma(c_, jx, s_ * jy);
          tx = fma(-s_, tx, c_ * ty_);
          jx = fma(-s_, jx, c_ * jy);
        }
        WSYNC();
      }
      WSYNC();
      if (dr) { if (s < q) T[s * QLD + q - 1] = 0.0; }
      WSYNC();
      if (dr) {
        if (s < q) T[(q - 1) * QLD + s] = 0.0;
        if (has_b) J[s * QLD + q - 1] = jx;
        --q;
      }
      WSYNC();
    };
    struct Zr { double z, rv, dq, jq; };
    auto products = [&](const bool want_r) -> Zr {
      Zr o;
      double z = 0.0, zb = 0.0, rv = 0.0, rvb = 0.0;
      o.dq = qdv[q & 15];
      o.jq = J[sJ * QLD + (q & 15)];
#pragma unroll
      for (int k = 0; k < QPV; k += 2) {
        const double2a j2 = lds2(J + sJ * QLD + k); const double2a y2 = lds2(qyv + k);
        z = fma(j2.x, y2.x, z); zb = fma(j2.y, y2.y, zb);
      }
      z += zb;
      if (want_r) {
#pragma unroll
        for (int k = 0; k < QPV; k += 2) {
          const double2a t2 = lds2(T + sT * QLD + k); const double2a d2_ = lds2(qdv + k);
          rv = fma(t2.x, d2_.x, rv); rvb = fma(t2.y, d2_.y, rvb);
        }
        rv += rvb;
      }
      if (s >= q) rv = 0.0;
      if (!has_b) z = 0.0;
      o.z = z; o.rv = rv;
      return o;
    };
    auto add_step = [&](const bool add, const double zn, const Zr& zr, const int wc, const double u_new) {
      const double rsz = frsq(zn), sz = zn * rsz;
      const double delta = (zr.dq >= 0.0) ? -sz : sz;
      const double hv = zn - delta * zr.dq;
      const double vv = 2.0 * hv;
      const double w = (zr.z - delta * zr.jq) * ((vv > 0.0) ? frcp(hv) : 0.0);
      if (add && has_b && vv > 0.0) {
#pragma unroll
        for (int k = 0; k < QPV; k += 2) {
          const double2a j2 = lds2(J + s * QLD + k); const double2a y2 = lds2(qyv + k);
          sts2(J + s * QLD + k, fma(-w, y2.x, j2.x), fma(-w, y2.y, j2.y));
        }
        J[s * QLD + q] = fma(-w, zr.dq - delta, zr.jq);
      }
      if (add) {
        const double idel = (zr.dq >= 0.0) ? -rsz : rsz;
        if (s < q) T[s * QLD + q] = -zr.rv * idel;
        if (s == q) { T[s * QLD + q] = idel; u = u_new; a_code = wc; }
        set_act(wc & 255, true);
        ++q;
      }
    };
    if (WARM) {
      // ================================ warm start (the packed sim3 kernel's scheme) ====================================================
      const unsigned long long ws0 = ((unsigned long long)(unsigned)bpermi((int)(ws_mine >> 32), rbase) << 32) | (unsigned)bpermi((int)(unsigned)ws_mine, rbase);
      const unsigned long long ws1 = ((unsigned long long)(unsigned)bpermi((int)(ws_mine >> 32), rbase + 1) << 32) | (unsigned)bpermi((int)(unsigned)ws_mine, rbase + 1);
      auto bits = [](const unsigned long long w, const int i) -> int { return (int)(((w >> (i & 31)) & 1ull) | (((w >> (32 + (i & 31))) & 1ull) << 1)); };
      const int tbase = c_con_com ? 2 : 0;   // findConstraints' order: CoM rows, then the trunk box
      const int dofv = P.q_red2dof[s & 15], dofr0 = P.q_bl2dof[(s >= 6 ? s - 6 : 0) % 18], dofr1 = P.q_bl2dof[(10 + s) % 18];
      int sb = (has_b && s >= 6) ? bits(ws0, dofv) : 0;
      int sr0 = (s < 4) ? (c_con_trunk ? bits(ws1, tbase + s) : 0) : ((s < 6) ? (c_con_com ? bits(ws1, s - 4) : 0) : bits(ws0, dofr0));
      int sr1 = has_r1 ? bits(ws0, dofr1) : 0;
      if (sb == 3) sb = 0;
      if (sr0 == 3) sr0 = 0;
      if (sr1 == 3) sr1 = 0;
      const double x0r = x;
      WSYNC();
      qxv[s] = has_b ? x : 0.0;
      WSYNC();
      const double near = 0.25 * fmax(1.0, -rmin16(has_b ? -fabs(x) : 0.0));
      const double v0 = row_dot(s, qxv), v1 = has_r1 ? row_dot(16 + s, qxv) : 0.0;
      const double slb = (sb == 2) ? ub - x : x - lb;
      const double sl0 = (sr0 == 2) ? cub0 - v0 : v0 - clb0;
      const double sl1 = (sr1 == 2) ? cub1 - v1 : v1 - clb1;
      bool pend_b = searching && has_b && ((sb == 1 && lb > -QP_INF) || (sb == 2 && ub < QP_INF)) && (slb <= near);
      bool pend_0 = searching && ((sr0 == 1 && clb0 > -QP_INF) || (sr0 == 2 && cub0 < QP_INF)) && (sl0 <= near);
      bool pend_1 = searching && has_r1 && ((sr1 == 1 && clb1 > -QP_INF) || (sr1 == 2 && cub1 < QP_INF)) && (sl1 <= near);
      bool seeded = false;
#pragma unroll 1
      for (;;) {                            // one seed per row and pass: bounds first, then rows 0..15, then rows 16..23; lowest index first
        const unsigned mb = (unsigned)((__ballot(pend_b) >> rbase) & 0xFFFFull), m0 = (unsigned)((__ballot(pend_0) >> rbase) & 0xFFFFull),
                       m1 = (unsigned)((__ballot(pend_1) >> rbase) & 0xFFFFull);
        const bool seeding = (mb | m0 | m1) != 0u;
        if (!__ballot(seeding)) break;
        const int kind = mb ? 0 : (m0 ? 1 : 2);
        const int idx = seeding ? __ffs((int)(kind == 0 ? mb : (kind == 1 ? m0 : m1))) - 1 : 0;
        if (seeding && s == idx) { if (kind == 0) pend_b = false; else if (kind == 1) pend_0 = false; else pend_1 = false; }
        const int my_side = ((kind == 0 ? sb : (kind == 1 ? sr0 : sr1)) == 2) ? 256 : 0;
        const double my_n2 = (kind == 0) ? 1.0 : ((kind == 1) ? cn0 : cn1);
        const int wsrc = rbase + idx;
        const int wc = ((kind == 0 ? idx : 32 + 16 * (kind - 1) + idx) & 255) | bpermi(my_side, wsrc);
        const double np2 = bperm(my_n2, wsrc);
        const int ip = wc & 255;
        const bool is_row = ip >= 32;
        const int rr_ = is_row ? ip - 32 : 0;
        const double sgn = (wc >> 8) ? -1.0 : 1.0;
        double d = normal_d(is_row, rr_, ip, sgn);
        if (!has_b || !seeding) d = 0.0;
        WSYNC();
        qdv[s] = d; qyv[s] = (s >= q) ? d : 0.0;
        WSYNC();
        const double zn = rsum16(s >= q ? d * d : 0.0);
        const Zr zr = products(__ballot(seeding && q > 0) != 0);
        const bool add = seeding && (zn > 100.0 * n * EPS2 * jf2 * np2) && q < QTC;      // (a dependent seed, or one more than T holds, is simply not taken)
        if (__ballot(add)) {
          add_step(add, zn, zr, wc, 0.0);
          if (add) { seeded = true; ++iters; }
        }
      }
      // x, u from the factors: with s_j = b_j - n_j'x0 the slacks of the slots at x0:  w = T's,  x = x0 + J1 w,  u = T w
      auto refresh = [&](const bool on) {
        const int cc = a_code & 255;
        const int rr = cc >= 32 ? cc - 32 : 0;
        const double f_b = bperm(-slb, rbase + (cc & 15)), f_0 = bperm(-sl0, rbase + (rr & 15)), f_1 = bperm(-sl1, rbase + (rr & 15));   // (every lane takes part)
        const double sj = (s < q) ? ((cc < 32) ? f_b : ((rr < 16) ? f_0 : f_1)) : 0.0;
        WSYNC();
        qdv[s] = sj;
        WSYNC();
        double w = 0.0;
#pragma unroll
        for (int j = 0; j < QTC; ++j) w = fma(T[j * QLD + sT], qdv[j], w);
        WSYNC();
        qyv[s] = (s < q && s < QTC) ? w : 0.0;
        WSYNC();
        double xa = 0.0, ua = 0.0;
#pragma unroll
        for (int kk = 0; kk < QPV; kk += 2) {
          const double2a j2 = lds2(J + sJ * QLD + kk), t2 = lds2(T + sT * QLD + kk), w2 = lds2(qyv + kk);
          xa = fma(j2.x, w2.x, fma(j2.y, w2.y, xa)); ua = fma(t2.x, w2.x, fma(t2.y, w2.y, ua));
        }
        if (on) { x = has_b ? x0r + xa : 0.0; u = (s < q) ? ua : 0.0; }
      };
      if (__ballot(seeded)) {
        refresh(seeded);
        bool restoring = seeded, did = false, again = false;
#pragma unroll 1
        for (;;) {                          // RESTORATION: while a seeded multiplier is negative the most negative slot is dropped
          const double um = rmin16((s < q) ? u : 0.0);
          bool rest = restoring && (um < 0.0);
          if (rest && ++iters > max_iter) { status = WBC_QP_MAX_ITER; rest = false; restoring = false; searching = false; }
          if (!__ballot(rest)) {
            if (!__ballot(restoring && did && !again)) break;
            const bool on = restoring && did && !again;
            refresh(on);
            if (on) again = true;
            continue;
          }
          const int l = rest ? __ffs((int)((__ballot(rest && s < q && u == um) >> rbase) & 0xFFFFull)) - 1 : 0;
          const int lcode = bpermi(a_code, rbase + (l < 0 ? 0 : l));
          drop_slot(rest, l < 0 ? 0 : l);
          const int ip = lcode & 255;
          const bool is_row = ip >= 32;
          const int rr_ = is_row ? ip - 32 : 0;
          double d = normal_d(is_row, rr_, ip, (lcode >> 8) ? -1.0 : 1.0);
          if (!has_b || !rest) d = 0.0;
          WSYNC();
          qdv[s] = d; qyv[s] = (s >= q) ? d : 0.0;
          WSYNC();
          const Zr zr = products(__ballot(rest && q > 0) != 0);
          if (rest) { x = fma(-um, zr.z, x); u = fma(um, zr.rv, u); did = true; }
        }
      }
    }
#pragma unroll 1
    for (;;) {
      WSYNC();
      qxv[s] = has_b ? x : 0.0;
      WSYNC();
      double best = 0.0; int code = -1;
      double cand_b = 0.0, cand_n2 = 1.0;
      if (has_b && !(actm & 1)) {
        if (lb > -QP_INF) { const double sl = x - lb; if (sl < -1e-9 * fmax(1.0, fabs(lb)) && sl < best) { best = sl; code = s; cand_b = lb; } }
        if (ub < QP_INF) { const double sl = ub - x; if (sl < -1e-9 * fmax(1.0, fabs(ub)) && sl < best) { best = sl; code = s | 256; cand_b = -ub; } }
      }
      if (!(actm & 2)) {
        const double v = row_dot(s, qxv);
        if (clb0 > -QP_INF) { const double sl = v - clb0; if (sl < -1e-9 * fmax(1.0, fabs(clb0)) && sl < best) { best = sl; code = 32 + s; cand_b = clb0; cand_n2 = cn0; } }
        if (cub0 < QP_INF) { const double sl = cub0 - v; if (sl < -1e-9 * fmax(1.0, fabs(cub0)) && sl < best) { best = sl; code = (32 + s) | 256; cand_b = -cub0; cand_n2 = cn0; } }
      }
      if (has_r1 && !(actm & 4)) {
        const double v = row_dot(16 + s, qxv);
        if (clb1 > -QP_INF) { const double sl = v - clb1; if (sl < -1e-9 * fmax(1.0, fabs(clb1)) && sl < best) { best = sl; code = 48 + s; cand_b = clb1; cand_n2 = cn1; } }
        if (cub1 < QP_INF) { const double sl = cub1 - v; if (sl < -1e-9 * fmax(1.0, fabs(cub1)) && sl < best) { best = sl; code = (48 + s) | 256; cand_b = -cub1; cand_n2 = cn1; } }
      }
      const double worst = rmin16(best);
      if (searching && !(worst < 0.0)) searching = false;
      if (!__ballot(searching)) break;
      const unsigned long long wm = __ballot(searching && best == worst);
      const int wl = __ffs((int)((wm >> rbase) & 0xFFFFull)) - 1;
      const int wsrc = rbase + (wl < 0 ? 0 : wl);
      const int wc = bpermi(code, wsrc);
      const double b_ip = bperm(cand_b, wsrc);
      const double np2 = bperm(cand_n2, wsrc);
      const int ip = wc & 255, ip_side = (wc >> 8) & 1;
      const bool is_row = ip >= 32;
      const int rr_ = is_row ? ip - 32 : 0;
      const double sgn = ip_side ? -1.0 : 1.0;
      double s_ip = worst, u_ip = 0.0;
      bool stepping = searching;
      int drop_l = -1;
#pragma unroll 1
      for (;;) {
        if (stepping && ++iters > max_iter) { status = WBC_QP_MAX_ITER; stepping = false; searching = false; }
        if (__ballot(stepping && drop_l >= 0)) {
          const bool dr = stepping && drop_l >= 0;
          drop_slot(dr, drop_l);
          qxv[s] = has_b ? x : 0.0;
          WSYNC();
          const double v = is_row ? row_dot(rr_, qxv) : qxv[ip & 15];
          if (dr) { s_ip = sgn * v - b_ip; drop_l = -1; }
        }
        if (!__ballot(stepping)) break;
        double d = normal_d(is_row, rr_, ip, sgn);
        if (!has_b || !stepping) d = 0.0;
        WSYNC();
        qdv[s] = d; qyv[s] = (s >= q) ? d : 0.0;
        WSYNC();
        const double zn = rsum16(s >= q ? d * d : 0.0);
        const Zr zr = products(__ballot(stepping && q > 0) != 0);
        const double z = zr.z, rv = zr.rv;
        const bool have_step = zn > 100.0 * n * EPS2 * jf2 * np2;
        const bool cand = (s < q) && (rv > 2.2250738585072014e-308);
        const double ratio = cand ? u * frcp(rv) : INFINITY;
        const double t1 = rmin16(ratio);
        const unsigned long long lm = __ballot(cand && ratio == t1);
        const int l = (t1 < INFINITY) ? __ffs((int)((lm >> rbase) & 0xFFFFull)) - 1 : -1;
        const double t2 = have_step ? -s_ip * frcp(zn) : INFINITY;
        const double tt = fmin(t1, t2);
        if (stepping && !(tt < INFINITY)) { status = WBC_QP_INFEASIBLE; stepping = false; searching = false; }
        if (stepping) {
          if (have_step) x = fma(tt, z, x);
          u = fma(-tt, rv, u);
          u_ip += tt;
        }
        bool add = stepping && have_step && tt == t2;
        if (add && q >= QTC) { overflow = true; add = false; stepping = false; searching = false; }
        if (__ballot(add)) {
          add_step(add, zn, zr, wc, u_ip);
          if (add) stepping = false;
        }
        if (stepping) drop_l = l;
      }
    }
    {   // more active constraints than T holds: the instance goes to the tail with the flagged ones
      const unsigned long long om = __ballot(valid && overflow);
      if ((om >> rbase) & 0xFFFFull) { defer = true; live = false; }
    }
    if (WARM && A.ws_out) {   // the final working set in FULL-problem indexing; an unsolved QP carries nothing
      const int cc = a_code & 255, sd = (a_code >> 8) & 1;
      const int rr = cc >= 32 ? cc - 32 : 0;
      const int tbase = c_con_com ? 2 : 0;
      const int dv_ = P.q_red2dof[cc & 15], dr_ = P.q_bl2dof[(rr >= 6 ? rr - 6 : 0) % 18];
      if (status == WBC_QP_OPTIMAL && s < q) {
        if (cc < 32) ws_o0 = 1ull << (32 * sd + (dv_ & 31));
        else if (rr < 4) ws_o1 = 1ull << (32 * sd + tbase + rr);
        else if (rr < 6) ws_o1 = 1ull << (32 * sd + (rr - 4));
        else ws_o0 = 1ull << (32 * sd + (dr_ & 31));
      }
      ws_o0 = ror16(ws_o0); ws_o1 = ror16(ws_o1);
    }
  }
  if (status == WBC_QP_OPTIMAL) {
    const unsigned long long bad = __ballot(s < n && !(fabs(x) <= 1.7976931348623157e308));
    if ((bad >> rbase) & 0xFFFFull) status = WBC_QP_NUMERICAL;
  }
  if (status != WBC_QP_OPTIMAL) x = 0.0;
  // ---- qd = Z y by DoF, outputs
  I.xv[s] = x;
  WSYNC();
  {
    auto qd_of = [&](const DevPlan::QDof& D, const bool on) -> double {
      double v = 0.0;
      if (on && D.bl >= 0) {
        const double2a z0 = lds2(Zm + D.bl * 6), z1 = lds2(Zm + D.bl * 6 + 2), z2 = lds2(Zm + D.bl * 6 + 4);
        const double2a v0 = lds2(I.xv), v1 = lds2(I.xv + 2), v2 = lds2(I.xv + 4);
        v = fma(z0.x, v0.x, fma(z0.y, v0.y, fma(z1.x, v1.x, fma(z1.y, v1.y, fma(z2.x, v2.x, z2.y * v2.y)))));
      } else if (on && D.red >= 6) v = I.xv[D.red & 15];
      return v;
    };
    const double v0 = qd_of(D0, true), v1 = qd_of(D1, has1);
    I.gp[s] = v0; I.gp[16 + s] = v1;
  }
  WSYNC();
  const bool wr = live;
  if (wr) {
    double* qo = A.out.qdot + (size_t)b * NV;
    qo[s] = I.gp[s];
    if (16 + s < NV) qo[16 + s] = I.gp[16 + s];
    if (s == 0) {
      A.out.status[b] = status;
      if (A.out.iters) A.out.iters[b] = nl + (INEQ ? iters + P.q_nlock : 0);     // (+ the eliminated equalities and the locked DoF, so that `iters` keeps its meaning)
    }
  }
  // working sets (a hot-started tick / roll-out of these configurations stays on this kernel): the problem has no inequality, so a carried set
  // seeds nothing and the set handed on is empty — as the general kernel reports it, tail instances included
  if (A.ws_out && valid && s < 2 && !(INEQ && WARM && defer)) A.ws_out[2 * (size_t)b + s] = (INEQ && WARM) ? (s == 0 ? ws_o0 : ws_o1) : 0ull;   // (a deferred instance's set is the tail's)
  if (A.out.q_next) {   // jointVelocitiestoConfig (Robot_Wrapper4.py:440-441)
    WSYNC();
    if (INEQ) {                              // (the QP's T took the staged configuration's place)
      const double* qg = A.in.q + (size_t)b * NQ;
      I.in[s] = qg[s];
      if (16 + s < 28) I.in[16 + s] = (16 + s < NQ) ? qg[16 + s] : 0.0;
    }
    I.xv[s] = (s < 6) ? I.gp[s] * dt : 0.0;
    WSYNC();
    double* qn = A.out.q_next + (size_t)b * NQ;
    if (wr) {
      integrate_ff(I, s, qn);
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int d = 6 + s + 16 * h2;
        if (d < nv) { const int qi = d + 1; qn[qi] = qv[qi] + I.gp[d] * dt; }   // (one free-flyer + 1-DoF joints: q index = DoF + 1)
      }
      if (s < NQ - nq) qn[nq + s] = 0.0;
    }
  }
  // ---- the tail: instances left out above (a flagged leg block; diagnostic orth_qr) are redone by this wave on the general path
  const unsigned long long tailm = __ballot(valid && defer && s == 0);
  if (tailm) {
    if (valid && defer && s == 0 && A.defer_stat) {
      unsigned long long old = *(volatile unsigned long long*)A.defer_stat, assumed;
      do {
        assumed = old;
        const unsigned long long cnt = ((assumed >> 32) == (unsigned long long)A.tick_seq) ? (assumed & 0xFFFFFFFFull) + 1ull : 1ull;
        old = atomicCAS(A.defer_stat, assumed, ((unsigned long long)A.tick_seq << 32) | cnt);
      } while (old != assumed);
    }
#pragma unroll 1
    for (int rr = 0; rr < 4; ++rr) {
      if (!((tailm >> (16 * rr)) & 1ull)) continue;
      if (INEQ && WARM) tail_instance<true, false>(&SU.G, 4 * (int)blockIdx.x + rr, models, cfgs, plans);     // (the general kernel's warm path: full-size solve)
      else tail_instance<false, true>(&SU.G, 4 * (int)blockIdx.x + rr, models, cfgs, plans);
    }
  }
}
template __global__ void wbc_tick_orthp_kernel<false>(const KernelArgs, const DevModel* __restrict__, const WbcConfig* __restrict__, const DevPlan* __restrict__);
template __global__ void wbc_tick_orthp_kernel<true>(const KernelArgs, const DevModel* __restrict__, const WbcConfig* __restrict__, const DevPlan* __restrict__);
template __global__ void wbc_tick_orthp_kernel<true, true>(const KernelArgs, const DevModel* __restrict__, const WbcConfig* __restrict__, const DevPlan* __restrict__);

// ================================================================================================
// The PACKED BOX kernel (round 3): FOUR instances per wavefront for the task problems WITHOUT constraint rows — the warm-up problem of
// setInitialState (Robot_Wrapper4.py:196-351: trunk + five EE tasks + posture, velocity box only; 2000 QPs per robot): n = 26 unknowns in
// 16 lanes. H = sum_t A_t'A_t + d^2 I is block-arrow: a limb's DoF meet the other limbs only through the six base DoF. The base — which the
// velocity box never holds on this controller — and, where 16 lanes do not hold the rest, the limb DoF with the widest position range (A1 +
// wx200: one thigh) are ELIMINATED by a Schur complement:
//     x = [x_E; x_K],   H_EE = L L',   W~ = L^-1 H_EK,   H' = H_KK - W~'W~,   g' = g_K - W~'(L^-1 g_E),   x_E = -L^-T (L^-1 g_E + W~ x_K)
// and the dual active-set method of the packed sim3 kernel (same lambdas, no general rows) solves  min 1/2 x_K'H'x_K + g'x_K, lb <= x_K <= ub
// on n' <= 16 bounded unknowns; DoF the box locks at 0 are left out. The dual iterates of the full problem ARE those of the reduced one
// (the eliminated unknowns are unconstrained minimisers at every step), so the working-set sequence and the iteration count are the
// oracle's; cond(H') <= cond(H), typically far below it (tests: 1e-12 against the oracle). The eliminated DoF's own velocity bounds are checked at the
// end: an instance that violates one (or needs more than XTC = 12 active bounds) is redone by its own wave on the general path (the tail).
// lane = 16 r + s: instance r; s = FK slot / DoF column s and 16 + s in the kinematics and task stage, eliminated slot s (< 8) and kept variable s
// from the Schur stage on. Stages: FK and columns as in the packed orth kernel; every task block's base columns -> Ab [task][row][8], the limb
// DoF's columns (each moves ONE task's frame) -> Ac [row][16]; H_EE rows on 8 lanes, cooperative 8 x 8 Cholesky; W~, H' rows, g' one kept
// variable per lane; two-column Cholesky sweep of H' fused with L y = e_s; dual iterations; x_E by eight row sums; outputs.
// ================================================================================================
constexpr int XLD = 18;                     // row stride of J (9 s mod 16 is a permutation: "lane = row" b128 reads are conflict-free)
constexpr int XTLD = 14, XTC = 12;          // T = R^-1: at most XTC active bounds, row stride XTLD
constexpr int DPP_ROR8 = 0x128;             // row_ror:8 — lane s <-> lane s ^ 8 of the 16-lane row
// R's length sets the distance between the four instances' blocks. 188 made it 5120 B = a multiple of the 256-byte bank row: every broadcast read
// (all lanes of an instance on one address, four instances on four) and every "lane = element" access of two instances then met in the same banks —
// 33 % of the LDS-active cycles were conflicts. 176 / 180 / 184 (5024 / 5056 / 5088 B: the instances 160 / 192 / 224 B apart mod 256) all measure
// 0.373 ms per 65536 ticks against 0.386.
constexpr int XRN = 176;
struct __attribute__((aligned(16))) XInst {
  double X[288];            // oMi [22][12] -> Ab [6][6][8]: base (+ eliminated limb DoF) columns of every task block -> W~ [8][16] -> J [16][XLD]
  double W[136];            // sin / cos [22][2] -> Ac [6][16] @0, g by DoF [32] @96 -> sweep vectors cl [32] @0, yv [32] @32 -> QP vectors xv @0, dv @16,
                            //   yv @32, tv @48 and, moved here from R before T is built, L [8][8] + 1 / L_jj [8] @64 -> qdot by DoF [32] @0
  double in[28];            // q [27]; WARM: the carried working set's bound word @27
  double R[XRN];            // ee_target [15] @0, prev_ee_target [15] @15, trunk inputs [18] @30, pf [16] @48, ow [16] @64, wt [96] @80
                            //   -> L [8][8] @0, 1 / L_jj [8] @64, g_E -> L^-1 g_E [8] @72 -> T [XTC][XTLD] @0, L^-1 g_E [8] @168
};
static_assert(sizeof(XInst) * 4 <= 20480, "8 waves per CU");
static_assert(XRN >= 176, "R holds the staged inputs [176] and T [168] + L^-1 g_E [8]");
struct XIntegrate { const double* in; const double* xv; };   // what integrate_ff reads

#ifdef WBC_ABLATE   // timing cuts 301.. (tools/ablate_boxp.py): the kernel returns after stage k with garbage
#define XSTOP(k, val) do { if (A.dbg_stop == 300 + (k)) { if (valid) { A.out.qdot[(size_t)b * NV + s] = (val); if (s == 0) A.out.status[b] = 0; } return; } } while (0)
#else
#define XSTOP(k, val) do { } while (0)
#endif
// WARM: the variant that takes / returns working sets (KernelArgs.ws_in / ws_out, word 0: velocity bounds by DoF) — the packed sim3 kernel's scheme
// (seeds through the add step, x / u rebuilt from the factors, restoration) on the kept variables; eliminated and locked DoF carry no seed.
template <bool WARM>
__global__ void __launch_bounds__(64, 2) wbc_tick_boxp_kernel(const KernelArgs A, const DevModel* __restrict__ models,
                                                              const WbcConfig* __restrict__ cfgs, const DevPlan* __restrict__ plans) {
  __shared__ union { XInst Q[4]; Smem G; } SU;
  const int lane = threadIdx.x, r = lane >> 4, s = lane & 15, rbase = lane & 48;
  XInst& I = SU.Q[r];
  const int b_raw = 4 * blockIdx.x + r;
  const bool valid = b_raw < A.B;
  const int b = valid ? b_raw : A.B - 1;
  int mid = 0;
  if (A.in.model_id) { mid = A.in.model_id[b]; mid = mid < 0 ? 0 : (mid >= A.n_models ? A.n_models - 1 : mid); }
  const DevModel& M = models[mid];
  const WbcConfig& cfg = cfgs[mid];
  const DevPlan& P = plans[mid];
  const double dt = A.dt, inv_dt = 1.0 / A.dt;
  double* const et = I.R;                    // ee_target [5][3]
  double* const ep = I.R + 15;               // prev_ee_target [5][3]
  double* const tin = I.R + 30;              // trunk_target [3], prev_trunk_target [3], trunk_ref_euler [3], trunk_prev_rot [9] -> target velocity x trunk_w [6]
  double* const pf = I.R + 48;               // EE frame origins [5][3]
  double* const ow = I.R + 64;               // the EE tasks' reference angular velocities [5][3]
  double* const wt = I.R + 80;               // the configuration's task weights and gains (WT_* offsets; trunk_W [6] @65, trunk_w @71, trunk_gain [6] @72, joint_w @84)
  const bool c_trunk = (P.flags & 4u) != 0;
  // ---- loads: inputs (coalesced per instance), then the per-lane records
  {
    const double* qg = A.in.q + (size_t)b * NQ;
    const double q0 = qg[s], q1 = (16 + s < NQ) ? qg[16 + s] : 0.0;
    const double e_t = (s < 15 && A.in.ee_target) ? A.in.ee_target[(size_t)b * 15 + s] : 0.0;
    const double e_p = (s < 15 && A.in.prev_ee_target) ? A.in.prev_ee_target[(size_t)b * 15 + s] : 0.0;
    I.in[s] = q0;
    if (16 + s < (WARM ? 27 : 28)) I.in[16 + s] = q1;
    if (WARM && s == 0) {                   // the carried working set's bound word, parked (as a bit pattern) in in[27]
      const unsigned long long w = (A.ws_in && valid) ? A.ws_in[2 * (size_t)b] : 0ull;
      I.in[27] = __longlong_as_double((long long)w);
    }
    if (s < 15) { et[s] = e_t; ep[s] = e_p; }
    if (c_trunk) {
      auto tinv = [&](const int k) -> double {
        return (k < 3) ? A.in.trunk_target[(size_t)b * 3 + k] : (k < 6) ? A.in.prev_trunk_target[(size_t)b * 3 + (k - 3)]
             : (k < 9) ? A.in.trunk_ref_euler[(size_t)b * 3 + (k - 6)] : A.in.trunk_prev_rot[(size_t)b * 9 + (k - 9)];
      };
      const double t0 = tinv(s), t1 = (s < 2) ? tinv(16 + s) : 0.0;
      tin[s] = t0;
      if (s < 2) tin[16 + s] = t1;
    }
    {   // calcTargetVelEE3's orientation feed-forward (Robot_Wrapper4.py:1125-1133), one component per lane (as in the packed orth kernel)
      double om = 0.0;
      if (A.in.ee_ref_rot && s < 15) {
        const int e = s / 3, i = s - 3 * e;
        const double* Rs = A.in.ee_ref_rot + (size_t)b * 45 + 9 * e;
        const double* Rp = A.in.ee_prev_rot + (size_t)b * 45 + 9 * e;
        const int ra = (i == 0) ? 6 : ((i == 1) ? 0 : 3), rb = (i == 0) ? 3 : ((i == 1) ? 6 : 0);
        om = ((Rs[ra] - Rp[ra]) * inv_dt) * Rs[rb] + ((Rs[ra + 1] - Rp[ra + 1]) * inv_dt) * Rs[rb + 1] + ((Rs[ra + 2] - Rp[ra + 2]) * inv_dt) * Rs[rb + 2];
      }
      ow[s] = om;
    }
    const double* cw = &cfg.ee_W[0][0];
#pragma unroll
    for (int i = 0; i < 6; ++i) wt[s + 16 * i] = (s + 16 * i < 85) ? cw[s + 16 * i] : 0.0;
  }
  const int nv = M.nv, nq = M.nq, nk = P.x_nk, ne = P.x_ne;
  const DevPlan::QDof D0 = P.q_dof[s], D1 = P.q_dof[16 + s];
  const int role0 = P.x_role[s], role1 = P.x_role[16 + s];
  const DevPlan::XVar kv = P.x_kept[s], ev = P.x_elim[s & 7];
  const unsigned limb = P.x_limb[s];
  DevPlan::PkJoint fkn = P.q_fk[0][s];
  const int scq0 = P.q_scq[(2 + s) & 31], scq1 = P.q_scq[(18 + s) & 31];
  const bool has1 = 16 + s < nv;
  const int c_task_joint = cfg.task_joint;
  const double dcoef = cfg.damper_coef, dqi = cfg.damper_qi, dqs = cfg.damper_qs;
  const int fjoint = (s < 5) ? M.frame_joint[WBC_FR_EE0 + s] : 1;
  const double fp0 = (s < 5) ? M.frame_p[WBC_FR_EE0 + s][0] : 0.0, fp1 = (s < 5) ? M.frame_p[WBC_FR_EE0 + s][1] : 0.0,
               fp2 = (s < 5) ? M.frame_p[WBC_FR_EE0 + s][2] : 0.0;
  WSYNC();
  const double* const qv = I.in;
  if (__ballot(c_trunk)) {
    // calcTargetVelTrunk2 (Robot_Wrapper4.py:948-1015) / TrunkB (:914-920), as in the packed sim3 kernel's TRUNK variant: the trunk frame is the
    // free-flyer's own placement (the plan checks it), so the target velocity depends on the inputs alone
    const double* tw = wt + 65;              // trunk_W [0..5], trunk_w [6], trunk_gain [7..12]
    const double* xt = tin;
    const double* xp = tin + 3;
    const double* er = tin + 6;
    double* const sh = I.X;                  // (free until the FK)
    double Rt_[9], fq[4], rq[4], Rs[9], vel[6];
    quat_to_R(qv + 3, Rt_);
    R_to_quat(Rt_, fq);
    {
      const SinCos t = sincos_cw(s < 3 ? er[s < 3 ? s : 0] : 0.5 * er[(s < 6 ? s : 3) - 3]);   // reference angles and their halves, one per lane
      if (s < 6) { sh[2 * s] = t.s; sh[2 * s + 1] = t.c; }
      WSYNC();
      const double sa = sh[0], ca = sh[1], sb = sh[2], cb = sh[3], sc_ = sh[4], cc = sh[5];
      Rs[0] = cc * cb; Rs[1] = cc * sb * sa - sc_ * ca; Rs[2] = cc * sb * ca + sc_ * sa;
      Rs[3] = sc_ * cb; Rs[4] = sc_ * sb * sa + cc * ca; Rs[5] = sc_ * sb * ca - cc * sa;
      Rs[6] = -sb;      Rs[7] = cb * sa;                 Rs[8] = cb * ca;
      const double qx[4] = {sh[6], 0, 0, sh[7]}, qy[4] = {0, sh[8], 0, sh[9]}, qz[4] = {0, 0, sh[10], sh[11]};
      double tq[4];
      quat_mul(qy, qx, tq);
      quat_mul(qz, tq, rq);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) vel[i] = (xt[i] - xp[i]) * inv_dt + tw[7 + i] * ((xt[i] - qv[i]) * inv_dt);
    const double qe0 = fq[3] * rq[0] - fq[0] * rq[3] + fq[1] * rq[2] - fq[2] * rq[1];   // :974
    const double qe1 = fq[3] * rq[1] - fq[1] * rq[3] - fq[0] * rq[2] + fq[2] * rq[0];   // :975
    const double qe2 = fq[3] * rq[2] - fq[3] * rq[2] + fq[0] * rq[1] - fq[1] * rq[0];   // :976 (sic)
    double D[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) D[i] = (Rs[i] - tin[9 + i]) * inv_dt;
    // skew = D Rs (R*, not R*^T: :984); omega = (S[2][1], S[0][2], S[1][0]) + K qe
    vel[3] = (D[6] * Rs[1] + D[7] * Rs[4] + D[8] * Rs[7]) + tw[10] * qe0;
    vel[4] = (D[0] * Rs[2] + D[1] * Rs[5] + D[2] * Rs[8]) + tw[11] * qe1;
    vel[5] = (D[3] * Rs[0] + D[4] * Rs[3] + D[5] * Rs[6]) + tw[12] * qe2;
    const double trunk_w = tw[6];
    WSYNC();                                 // (everyone has read the inputs)
    if (s == 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i) tin[i] = vel[i] * trunk_w;
    }
    WSYNC();
  }
  double* const oMi = I.X;                   // [22][12]
  double* const sc = I.W;                    // sin / cos of joint j at 2 j
  {
    if (scq0 >= 0) { const SinCos t = sincos_cw(qv[scq0]); sc[2 * (2 + s)] = t.s; sc[2 * (2 + s) + 1] = t.c; }
    if (scq1 >= 0) { const SinCos t = sincos_cw(qv[scq1]); sc[2 * (18 + s)] = t.s; sc[2 * (18 + s) + 1] = t.c; }
    if (s == 0) {   // root free-flyer (joint 1): R from the quaternion as Eigen's toRotationMatrix, p = xyz; R column-major then p
      double Rt[9];
      quat_to_R(qv + 3, Rt);
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) oMi[12 + 3 * c + rr] = Rt[3 * rr + c];
      oMi[12 + 9] = qv[0]; oMi[12 + 10] = qv[1]; oMi[12 + 11] = qv[2];
    }
  }
  WSYNC();
  XSTOP(1, sc[4 + s] + oMi[12 + s] + kv.d_lo + ev.d_hi);
  // ---- pin.forwardKinematics, level by level (Robot_Wrapper4.py:400)
#pragma unroll 1
  for (int L = 0; L < QLEV; ++L) {
    const DevPlan::PkJoint fk = fkn;
    if (L + 1 < QLEV) fkn = P.q_fk[L + 1][s];
    const int j = fk.joint;
    if (j >= 0) {
      const bool rev = fk.rev != 0;
      const int a0 = fk.a0, a1 = fk.a1, a2 = fk.a2;
      const double* Pp = oMi + 12 * fk.parent;
      const double sn = rev ? sc[2 * j] : 0.0, cs = rev ? sc[2 * j + 1] : 1.0;
      const double pris = rev ? 0.0 : qv[fk.q_idx];
      double Av[3], Bv[3], Cv[3], Pv[3];
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) { Av[rr] = Pp[a0 + rr]; Bv[rr] = Pp[a1 + rr]; Cv[rr] = Pp[a2 + rr]; Pv[rr] = Pp[9 + rr]; }
      double* Po = oMi + 12 * j;
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        Po[a0 + rr] = Av[rr];
        Po[a1 + rr] = cs * Bv[rr] + sn * Cv[rr];
        Po[a2 + rr] = cs * Cv[rr] - sn * Bv[rr];
        Po[9 + rr] = Pv[rr] + Av[rr] * (fk.t0 + pris) + Bv[rr] * fk.t1 + Cv[rr] * fk.t2;
      }
    }
    WSYNC();
  }
  XSTOP(2, oMi[12 * 4 + (s & 7)]);
  // ---- frame origins (updateFramePlacements, :405), Jacobian columns (WORLD) of DoF s and 16 + s
  if (s < 5) {
    const double* Pg = oMi + 12 * fjoint;
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) pf[3 * s + rr] = Pg[9 + rr] + Pg[rr] * fp0 + Pg[3 + rr] * fp1 + Pg[6 + rr] * fp2;
  }
  double lin0[3] = {0, 0, 0}, ang0[3] = {0, 0, 0}, lin1[3] = {0, 0, 0}, ang1[3] = {0, 0, 0};
  {
    const double* Pj = oMi + 12 * D0.joint;
    const double pj[3] = {Pj[9], Pj[10], Pj[11]};
    if (D0.ang >= 0) { ang0[0] = Pj[3 * D0.ang]; ang0[1] = Pj[3 * D0.ang + 1]; ang0[2] = Pj[3 * D0.ang + 2]; cross3(pj, ang0, lin0); }
    if (D0.lin >= 0) { lin0[0] = Pj[3 * D0.lin]; lin0[1] = Pj[3 * D0.lin + 1]; lin0[2] = Pj[3 * D0.lin + 2]; }
  }
  if (has1) {
    const double* Pj = oMi + 12 * D1.joint;
    const double pj[3] = {Pj[9], Pj[10], Pj[11]};
    if (D1.ang >= 0) { ang1[0] = Pj[3 * D1.ang]; ang1[1] = Pj[3 * D1.ang + 1]; ang1[2] = Pj[3 * D1.ang + 2]; cross3(pj, ang1, lin1); }
    if (D1.lin >= 0) { lin1[0] = Pj[3 * D1.lin]; lin1[1] = Pj[3 * D1.lin + 1]; lin1[2] = Pj[3 * D1.lin + 2]; }
  }
  // ---- velDamperJointConstraints (:572-637): of kept variable s and of eliminated DoF s (< ne; checked at the end)
  double lb = 0.0, ub = 0.0, elb = 0.0, eub = 0.0;
  {
    auto damper = [&](const double qi, const double lo, const double hi, const double vm, double& l_, double& u_) {
      if (qi <= lo + dqi) { l_ = -dcoef * (qi - lo - dqs) / (dqi - dqs); if (l_ > vm) l_ = vm; if (l_ < -vm) l_ = -vm; } else l_ = -vm;
      if (qi >= hi - dqi) { u_ = dcoef * (hi - qi - dqs) / (dqi - dqs); if (u_ < -vm) u_ = -vm; if (u_ > vm) u_ = vm; } else u_ = vm;
      if (l_ > 0) l_ = -l_;
      if (u_ < 0) u_ = -u_;
    };
    if (s < nk) damper(qv[kv.dq_idx], kv.d_lo, kv.d_hi, kv.d_vm, lb, ub);
    if (s < ne) damper(qv[ev.dq_idx], ev.d_lo, ev.d_hi, ev.d_vm, elb, eub);
  }
  WSYNC();   // oMi is dead: X is free
  XSTOP(3, lin0[0] + ang0[1] + lin1[2] + ang1[0] + lb + eub);
  // ---- the task stack (qpA / qpb, Robot_Wrapper4.py:1271-1294): block t = 0 trunk, 1 + e EE e. Weighted columns of the eliminated DoF -> Ab [t][row][slot],
  // of the kept variables -> Ac [row][variable] (a limb DoF moves one task's frame: one column each); g by DoF in registers
  double* const Ab = I.X;                    // [6][6][8]
  double* const Ac = I.W;                    // [6][16]
  double* const gd = I.W + 96;               // [32]
  {
#pragma unroll
    for (int i = 0; i < 9; ++i) sts2(Ab + 2 * (s + 16 * i), 0.0, 0.0);
#pragma unroll
    for (int i = 0; i < 6; ++i) Ac[s + 16 * i] = 0.0;
  }
  const double joint_w = wt[84];
  const double dpost = (1.0 / nv) * joint_w;
  double g0 = 0.0, g1 = 0.0;                 // g of DoF s / 16 + s
  if (c_task_joint == WBC_JOINT_PREV) {      // qpJointb "PREV" (:1199-1268): q as the velocity target
    g0 = -dpost * ((1.0 / nv) * qv[s < 6 ? s : s + 1] * joint_w);
    g1 = has1 ? -dpost * ((1.0 / nv) * qv[17 + s] * joint_w) : 0.0;
  }
  WSYNC();
  auto emit = [&](const int t, const bool sup0, const bool sup1, const double* a0, const double* a1, const double* br) {
    g0 = fma(-a0[0], br[0], fma(-a0[1], br[1], fma(-a0[2], br[2], fma(-a0[3], br[3], fma(-a0[4], br[4], fma(-a0[5], br[5], g0))))));
    g1 = fma(-a1[0], br[0], fma(-a1[1], br[1], fma(-a1[2], br[2], fma(-a1[3], br[3], fma(-a1[4], br[4], fma(-a1[5], br[5], g1))))));
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) {
      if (sup0) { if (role0 >= 16) Ac[rr * 16 + (role0 - 16)] = a0[rr]; else if (role0 >= 0) Ab[(t * 6 + rr) * 8 + role0] = a0[rr]; }
      if (sup1) { if (role1 >= 16) Ac[rr * 16 + (role1 - 16)] = a1[rr]; else if (role1 >= 0) Ab[(t * 6 + rr) * 8 + role1] = a1[rr]; }
    }
  };
  if (__ballot(c_trunk)) {   // trunkA (Robot_Wrapper4.py:487-490, WORLD): support = the base DoF
    const double* tw = wt + 65;
    const double trunk_w = tw[6];
    const bool sup = c_trunk && s < 6;
    double at[6], a1[6] = {0, 0, 0, 0, 0, 0}, br[6];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
      at[rr] = sup ? (tw[rr] * lin0[rr]) * trunk_w : 0.0;
      at[3 + rr] = sup ? (tw[3 + rr] * ang0[rr]) * trunk_w : 0.0;
    }
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) br[rr] = c_trunk ? tin[rr] : 0.0;
    emit(0, sup, false, at, a1, br);
  }
  const unsigned tmask = (unsigned)__builtin_amdgcn_readfirstlane((int)P.task_ee_mask);
#pragma unroll 1
  for (unsigned tm = tmask; tm; tm &= tm - 1) {   // endEffectorA2 (:474-484) / calcTargetVelEE3 (:1052-1157) / EndEffectorB2 (:907-910)
    const int e = __ffs((int)tm) - 1;
    const double w = wt[WT_w + e];
    double Wd[6], Gd[3];
#pragma unroll
    for (int i = 0; i < 6; ++i) Wd[i] = wt[WT_W + 6 * e + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) Gd[i] = wt[WT_G + 6 * e + i];
    const double pfe[3] = {pf[3 * e], pf[3 * e + 1], pf[3 * e + 2]};
    const bool sup0 = (D0.supmask >> e) & 1, sup1 = has1 && ((D1.supmask >> e) & 1);
    double a0[6], a1[6], br[6], wxp[3];
    cross3(ang0, pfe, wxp);
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) { a0[rr] = sup0 ? Wd[rr] * ((lin0[rr] + wxp[rr]) * w) : 0.0; a0[3 + rr] = sup0 ? Wd[3 + rr] * (ang0[rr] * w) : 0.0; }
    cross3(ang1, pfe, wxp);
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) { a1[rr] = sup1 ? Wd[rr] * ((lin1[rr] + wxp[rr]) * w) : 0.0; a1[3 + rr] = sup1 ? Wd[3 + rr] * (ang1[rr] * w) : 0.0; }
    const double* xt = et + 3 * e;
    const double* xp = ep + 3 * e;
#pragma unroll
    for (int i = 0; i < 3; ++i) { br[i] = ((xt[i] - xp[i]) * inv_dt + Gd[i] * ((xt[i] - pfe[i]) * inv_dt)) * w; br[3 + i] = ow[3 * e + i] * w; }
    emit(1 + e, sup0, sup1, a0, a1, br);
  }
  gd[s] = g0; gd[16 + s] = has1 ? g1 : 0.0;
  WSYNC();
  XSTOP(4, Ab[s] + Ac[s] + gd[s]);
  // ---- Schur stage. Lane e < 8: row e of H_EE (slots >= ne: identity rows); lane k: H_EK column k (-> w~ = L^-1 of it), the limb block of H_KK row k, g
  const double d2 = dpost * dpost;
  double Lr[8], wk[8], own[6], gk;
  {
    // (the 36 block rows are split between lane e and lane e + 8, the halves joined by one row rotation)
    const int e8 = s & 7, half = s >> 3;
#pragma unroll
    for (int k = 0; k < 8; ++k) Lr[k] = 0.0;
#pragma unroll 2
    for (int i = 0; i < 18; ++i) {
      const double* row = Ab + (18 * half + i) * 8;
      const double o_ = row[e8];
      const double2a v0 = lds2(row), v1 = lds2(row + 2), v2 = lds2(row + 4), v3 = lds2(row + 6);
      Lr[0] = fma(o_, v0.x, Lr[0]); Lr[1] = fma(o_, v0.y, Lr[1]); Lr[2] = fma(o_, v1.x, Lr[2]); Lr[3] = fma(o_, v1.y, Lr[3]);
      Lr[4] = fma(o_, v2.x, Lr[4]); Lr[5] = fma(o_, v2.y, Lr[5]); Lr[6] = fma(o_, v3.x, Lr[6]); Lr[7] = fma(o_, v3.y, Lr[7]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      Lr[k] += dpp<DPP_ROR8>(Lr[k]);
      if (k == e8) Lr[k] += d2;
      if (e8 >= ne) Lr[k] = (k == e8) ? 1.0 : 0.0;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) own[rr] = (s < nk) ? Ac[rr * 16 + s] : 0.0;
    const double* Abt = Ab + (1 + (kv.task < 0 ? 0 : kv.task)) * 48;   // the kept variable's task block (a variable no task moves: own = 0)
#pragma unroll
    for (int k = 0; k < 8; ++k) wk[k] = 0.0;
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) {
      const double2a v0 = lds2(Abt + rr * 8), v1 = lds2(Abt + rr * 8 + 2), v2 = lds2(Abt + rr * 8 + 4), v3 = lds2(Abt + rr * 8 + 6);
      wk[0] = fma(own[rr], v0.x, wk[0]); wk[1] = fma(own[rr], v0.y, wk[1]); wk[2] = fma(own[rr], v1.x, wk[2]); wk[3] = fma(own[rr], v1.y, wk[3]);
      wk[4] = fma(own[rr], v2.x, wk[4]); wk[5] = fma(own[rr], v2.y, wk[5]); wk[6] = fma(own[rr], v3.x, wk[6]); wk[7] = fma(own[rr], v3.y, wk[7]);
    }
    gk = (s < nk) ? gd[kv.dof & 31] : 0.0;
  }
  const double ge_own = (s < ne) ? gd[ev.dof & 31] : 0.0;
  WSYNC();                                   // (Ab and the inputs have been read: X and R are free; Ac stays for the limb blocks)
  double* const Lq = I.R;                    // finished rows of L [8][8], 1 / L_jj at [64 + j]; g_E -> L^-1 g_E at [72 + j]
  double* const Wt = I.X;                    // W~ [8][16]
  if (s < 8) Lq[72 + s] = ge_own;
  __builtin_amdgcn_sched_barrier(0);
  // cooperative Cholesky H_EE = L L' (lane e keeps row e; a finished row and the reciprocal of its pivot go through LDS — as in the packed orth
  // kernel) with the forward substitutions riding on it: step j's row of L also finishes entry j of w~ = L^-1 (H_EK column) and of gt = L^-1 g_E
  {
    double gt[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      double v = Lr[j];
#pragma unroll
      for (int k = 0; k < j; ++k) v = fma(-Lr[k], Lr[k], v);
      v = (s == j) ? v : 1.0;
      double rs = __builtin_amdgcn_rsq(v);
      rs = rs * fma(-0.5 * v * rs, rs, 1.5); rs = rs * fma(-0.5 * v * rs, rs, 1.5);
      if (s == j) {
        Lr[j] = v * rs;
#pragma unroll
        for (int k = 0; k <= j; ++k) Lq[j * 8 + k] = Lr[k];
        Lq[64 + j] = (v > 0.0) ? rs : __builtin_nan("");      // (a failed pivot poisons everything downstream: status "numerical")
      }
      WSYNC();
      double w = Lr[j], t = wk[j], u_ = Lq[72 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) { const double l = Lq[j * 8 + k]; w = fma(-Lr[k], l, w); t = fma(-wk[k], l, t); u_ = fma(-gt[k], l, u_); }
      const double ri = Lq[64 + j];
      if (s > j && s < 8) Lr[j] = w * ri;
      wk[j] = t * ri; gt[j] = u_ * ri;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) gk = fma(-wk[k], gt[k], gk);
    WSYNC();                                 // (everyone has read g_E)
    if (s == 0) {
#pragma unroll
      for (int k = 0; k < 8; k += 2) sts2(Lq + 72 + k, gt[k], gt[k + 1]);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) Wt[k * 16 + s] = (s < nk) ? wk[k] : 0.0;
  WSYNC();
  __builtin_amdgcn_sched_barrier(0);
  // row s of H' = (limb block of H_KK) - W~'W~
  double h[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) h[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 16; k += 2) {
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) { const double2a v = lds2(Ac + rr * 16 + k); h[k] = fma(own[rr], v.x, h[k]); h[k + 1] = fma(own[rr], v.y, h[k + 1]); }
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (!((limb >> k) & 1u)) h[k] = 0.0;
    if (k == s) h[k] += (s < nk) ? d2 : 1.0;
  }
#pragma unroll
  for (int k = 0; k < 16; k += 2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { const double2a v = lds2(Wt + i * 16 + k); h[k] = fma(-wk[i], v.x, h[k]); h[k + 1] = fma(-wk[i], v.y, h[k + 1]); }
  }
  if (s >= nk) {
#pragma unroll
    for (int k = 0; k < 16; ++k) h[k] = (k == s) ? 1.0 : 0.0;
    gk = 0.0;
  }
  double g = gk;
  XSTOP(5, h[0] + h[5] + h[15] + g + wk[0] + wk[7]);
  const bool has_b = s < nk;
  bool live = valid;
  int status = WBC_QP_OPTIMAL;
  if (live && ((has_b && ((lb != lb) || (ub != ub))) || (s < ne && ((elb != elb) || (eub != eub))))) status = WBC_QP_NUMERICAL;
  {
    const unsigned long long nb = __ballot(status != WBC_QP_OPTIMAL);
    if ((nb >> rbase) & 0xFFFFull) { status = WBC_QP_NUMERICAL; live = false; }
  }
  // ---- Cholesky H' = L L' fused with the substitution L y = e_s (two columns per trip: the packed kernels' sweep)
  double* const cl = I.W;                    // [32] (the sweep's first column vector; free afterwards: xv | dv)
  double* const yv = I.W + 32;               // [32] (yv | tv)
  double* const tv = I.W + 48;
  double* const xv = I.W;
  double* const dv = I.W + 16;
  double* const Lk = I.W + 64;               // L [8][8] and 1 / L_jj [8] move here for the solve (R becomes T); L^-1 g_E [8] at R [168..175]
  WSYNC();                                   // (Ac has been read)
  cl[16 + s] = 0.0; yv[16 + s] = 0.0;
  double y[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) y[k] = (k == s) ? 1.0 : 0.0;
  double pmin = 1.0;
#pragma unroll 1
  for (int j = 0; j < 16; j += 2) {
    WSYNC();
    cl[s] = h[0]; yv[s] = h[1];
    WSYNC();
    const double* c0 = cl + j;
    const double* c1 = yv + j;
    double cm0[16], cm1[16];
#pragma unroll
    for (int rr = 0; rr < 16; rr += 2) {
      const double2a v0 = lds2(c0 + rr), v1 = lds2(c1 + rr);
      cm0[rr] = v0.x; cm0[rr + 1] = v0.y; cm1[rr] = v1.x; cm1[rr + 1] = v1.y;
    }
    const double pj = cm0[0];
    pmin = (pj > 0.0) ? fmin(pmin, pj) : -1.0;
    const double rinv = rsqrt(pj), ipj = rinv * rinv;
    const double th = h[0] * ipj, ty = y[0] * ipj, yk = y[0] * rinv;
    const double h1 = fma(-th, cm0[1], h[1]), y1 = fma(-ty, cm0[1], y[1]);
    const double a = cm0[1];
#pragma unroll
    for (int rr = 1; rr < 16; ++rr) cm1[rr] = fma(-(cm0[rr] * ipj), a, cm1[rr]);
    const double pj2 = cm1[1];
    pmin = (pj2 > 0.0) ? fmin(pmin, pj2) : -1.0;
    const double rinv2 = rsqrt(pj2), ipj2 = rinv2 * rinv2;
    const double th2 = h1 * ipj2, ty2 = y1 * ipj2, yk2 = y1 * rinv2;
#pragma unroll
    for (int rr = 2; rr < 16; ++rr) h[rr - 2] = fma(-th2, cm1[rr], fma(-th, cm0[rr], h[rr]));
#pragma unroll
    for (int rr = 2; rr < 16; ++rr) y[rr - 2] = fma(-ty2, cm1[rr], fma(-ty, cm0[rr], y[rr]));
    y[14] = fma(-ty2, 0.0, yk); y[15] = yk2;
    h[14] = 0.0; h[15] = 0.0;
  }
  if (live && !(pmin > 0.0)) { status = WBC_QP_NUMERICAL; live = false; }
  XSTOP(6, y[0] + y[15] + h[0]);
  // y = row s of J0 = L^-T.  jf2 = |J0|_F^2 per instance
  double sq = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) sq = fma(y[k], y[k], sq);
  const double jf2 = rsum16(sq);
  double* const J = I.X;                     // [16][XLD]
  double* const T = I.R;                     // [XTC][XTLD]
  double* const gtv = I.R + 168;             // L^-1 g_E [8]
  WSYNC();
  {
    double mv[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) mv[i] = (s + 16 * i < 80) ? Lq[s + 16 * i] : 0.0;
    WSYNC();
#pragma unroll
    for (int i = 0; i < 5; ++i) { if (s + 16 * i < 72) Lk[s + 16 * i] = mv[i]; else if (s + 16 * i < 80) gtv[s + 16 * i - 72] = mv[i]; }
  }
#pragma unroll
  for (int k = 0; k < 16; k += 2) sts2(J + s * XLD + k, y[k], y[k + 1]);
  if (s < XTC) {
#pragma unroll
    for (int k = 0; k < XTLD; k += 2) sts2(T + s * XTLD + k, 0.0, 0.0);
  }
  tv[s] = g;
  WSYNC();
  // x0 = -J0 (J0' g'): the unconstrained minimiser
  double x;
  {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t = fma(J[i * XLD + s], tv[i], t);
    dv[s] = has_b ? -t : 0.0;
    WSYNC();
    double xa = 0.0, xb = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k += 2) { const double2a v2 = lds2(dv + k); xa = fma(y[k], v2.x, xa); xb = fma(y[k + 1], v2.y, xb); }
    x = has_b ? xa + xb : 0.0;
  }
  XSTOP(7, x + jf2);
  // ---- dual active-set iterations (the packed sim3 kernel's, bounds only; per-row state; loops run until every row of the wave is done)
  bool act_b = false, overflow = false;
  double u = 0.0;
  int a_code = 0, q = 0, iters = 0;
  const int max_iter = 10 * nv + 20;        // (the full problem's cap: n = nv unknowns, no rows — the oracle's count includes the locked DoF this kernel leaves out)
  bool searching = live;
  const int sT = s < XTC ? s : XTC - 1;      // (lanes beyond T's rows shadow its last row; they never write)
  // drop slot l of the rows `dr`: Givens sequence read off the removed row of T (rare path)
  auto drop_slot = [&](const bool dr, const int l_) {
    const int l = dr ? l_ : 0;
    const int lc = bpermi(a_code, rbase + l) & 255;
    if (dr && s == lc) act_b = false;
    WSYNC();
    yv[s] = u; tv[s] = (double)a_code;
    WSYNC();
    if (dr && s >= l && s < q - 1) { u = yv[s + 1]; a_code = (int)tv[s + 1]; }
    if (dr && s == q - 1) { u = 0.0; a_code = 0; }
    const int srow = (sT >= l) ? ((sT + 1 < XTC) ? sT + 1 : sT) : sT;
    double tx = T[srow * XTLD + l];
    double jx = J[s * XLD + l];
    double hrun = T[l * XTLD + l];
    const int kend = dr ? q - 1 : 0;    // this row's rotations run k = l .. q - 2
#pragma unroll 1
    for (int k0 = 0; k0 < XTC - 1; ++k0) {
      const bool on = dr && (l + k0 < kend);
      if (!__ballot(on)) break;
      const int k = on ? l + k0 : 0;
      const double tb = T[l * XTLD + k + 1];
      const double nrm2 = fma(hrun, hrun, tb * tb);
      double c_ = 1.0, s_ = 0.0, rho = 0.0;
      if (nrm2 > 0.0) { const double ri = rsqrt(nrm2); c_ = tb * ri; s_ = -hrun * ri; rho = nrm2 * ri; }
      const double ty_ = T[srow * XTLD + k + 1];
      const double jy = J[s * XLD + k + 1];
      WSYNC();
      if (on) {
        hrun = rho;
        if (s < q - 1) T[s * XTLD + k] = fma(c_, tx, s_ * ty_);
        if (has_b) J[s * XLD + k] = fma(c_, jx, s_ * jy);
        tx = fma(-s_, tx, c_ * ty_);
        jx = fma(-s_, jx, c_ * jy);
      }
      WSYNC();
    }
    WSYNC();
    if (dr) {
      if (s < q) T[s * XTLD + q - 1] = 0.0;
    }
    WSYNC();
    if (dr) {
      if (s < q) T[(q - 1) * XTLD + s] = 0.0;
      if (has_b) J[s * XLD + q - 1] = jx;
      --q;
    }
    WSYNC();
  };
  // with d staged (dv = d, yv = d restricted to the slots >= q): z = J2 d2, r = T d1, and the add step's dq = d_q, jq = J[s][q]
  struct Zr { double z, rv, dq, jq; };
  auto products = [&](const bool want_r) -> Zr {
    Zr o;
    double z = 0.0, zb = 0.0, rv = 0.0, rvb = 0.0;
    o.dq = dv[q & 15];
    o.jq = J[s * XLD + (q & 15)];
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      const double2a j2 = lds2(J + s * XLD + k); const double2a y2 = lds2(yv + k);
      z = fma(j2.x, y2.x, z); zb = fma(j2.y, y2.y, zb);
    }
    z += zb;
    if (want_r) {                           // r = T d1: nothing to do while no row of the wave holds an active bound
#pragma unroll
      for (int k = 0; k < XTC; k += 2) {
        const double2a t2 = lds2(T + sT * XTLD + k); const double2a d2_ = lds2(dv + k);
        rv = fma(t2.x, d2_.x, rv); rvb = fma(t2.y, d2_.y, rvb);
      }
      rv += rvb;
    }
    if (s >= q) rv = 0.0;
    if (!has_b) z = 0.0;
    o.z = z; o.rv = rv;
    return o;
  };
  // add: Householder P with P d2 = delta e1; J2 <- J2 P; T gets column (-r/delta, 1/delta); the new slot's multiplier is u_new
  auto add_step = [&](const bool add, const double zn, const Zr& zr, const int wc, const int ip, const double u_new) {
    const double rsz = frsq(zn), sz = zn * rsz;
    const double delta = (zr.dq >= 0.0) ? -sz : sz;
    const double hv = zn - delta * zr.dq;               // v'v / 2
    const double vv = 2.0 * hv;
    const double w = (zr.z - delta * zr.jq) * ((vv > 0.0) ? frcp(hv) : 0.0);
    if (add && has_b && vv > 0.0) {   // J2 <- J2 - w v', v = d2 - delta e_q (entry q stored with its own term after the sweep on d2)
#pragma unroll
      for (int k = 0; k < 16; k += 2) {
        const double2a j2 = lds2(J + s * XLD + k); const double2a y2 = lds2(yv + k);   // yv = d for k >= q, else 0
        sts2(J + s * XLD + k, fma(-w, y2.x, j2.x), fma(-w, y2.y, j2.y));
      }
      J[s * XLD + q] = fma(-w, zr.dq - delta, zr.jq);
    }
    if (add) {
      const double idel = (zr.dq >= 0.0) ? -rsz : rsz;
      if (s < q) T[s * XTLD + q] = -zr.rv * idel;
      if (s == q) { T[s * XTLD + q] = idel; u = u_new; a_code = wc; }
      if (s == (ip & 15)) act_b = true;
      ++q;
    }
  };
  if (WARM) {
    const unsigned long long ws0 = (unsigned long long)__double_as_longlong(I.in[27]);
    int sb = has_b ? (int)(((ws0 >> (kv.dof & 31)) & 1ull) | (((ws0 >> (32 + (kv.dof & 31))) & 1ull) << 1)) : 0;
    if (sb == 3) sb = 0;
    // a seed is taken only if the unconstrained minimiser x0 violates it or comes close to it (qp_core, solve_v3 `far`)
    const double x0r = x;
    const double near = 0.25 * fmax(1.0, -rmin16(has_b ? -fabs(x) : 0.0));
    const double slb = (sb == 2) ? ub - x : x - lb;      // slack of the seeded side at x0
    bool pend_b = live && has_b && ((sb == 1 && lb > -QP_INF) || (sb == 2 && ub < QP_INF)) && (slb <= near);
    bool seeded = false;
#pragma unroll 1
    for (;;) {                              // one seed per row and pass, lowest index first
      const unsigned mb = (unsigned)((__ballot(pend_b) >> rbase) & 0xFFFFull);
      const bool seeding = mb != 0u;
      if (!__ballot(seeding)) break;
      const int idx = seeding ? __ffs((int)mb) - 1 : 0;
      if (seeding && s == idx) pend_b = false;
      const int c_side = (sb == 2) ? 256 : 0;
      const int wc = (idx & 255) | bpermi(c_side, rbase + idx);
      const int ip = wc & 255;
      const double sgn = (wc >> 8) ? -1.0 : 1.0;
      double d = sgn * J[(ip & 15) * XLD + s];
      if (!has_b || !seeding) d = 0.0;
      WSYNC();
      dv[s] = d; yv[s] = (s >= q) ? d : 0.0;
      WSYNC();
      const double zn = rsum16(s >= q ? d * d : 0.0);
      const Zr zr = products(__ballot(seeding && q > 0) != 0);
      const bool add = seeding && (zn > 100.0 * nk * EPS2 * jf2) && q < XTC;      // (a dependent seed, or one more than T holds, is simply not taken)
      if (__ballot(add)) {
        add_step(add, zn, zr, wc, ip, 0.0);
        if (add) { seeded = true; ++iters; }
      }
    }
    // x, u from the factors: with s_j = b_j - n_j'x0 the slacks of the slots at x0:  w = T's,  x = x0 + J1 w,  u = T w
    auto refresh = [&](const bool on) {
      const int cc = a_code & 255;
      const double sv_ = bperm(-slb, rbase + (cc & 15));     // (every lane takes part: ds_bpermute reads nothing from a lane that is switched off)
      const double sj = (s < q) ? sv_ : 0.0;
      WSYNC();
      dv[s] = sj;
      WSYNC();
      double w = 0.0;
#pragma unroll
      for (int j = 0; j < XTC; ++j) w = fma(T[j * XTLD + sT], dv[j], w);        // column s of T (zero outside the slots)
      WSYNC();
      yv[s] = (s < q && s < XTC) ? w : 0.0;
      WSYNC();
      double xa = 0.0, ua = 0.0;
#pragma unroll
      for (int k = 0; k < 16; k += 2) { const double2a j2 = lds2(J + s * XLD + k), w2 = lds2(yv + k); xa = fma(j2.x, w2.x, fma(j2.y, w2.y, xa)); }
#pragma unroll
      for (int k = 0; k < XTC; k += 2) { const double2a t2 = lds2(T + sT * XTLD + k), w2 = lds2(yv + k); ua = fma(t2.x, w2.x, fma(t2.y, w2.y, ua)); }
      if (on) { x = has_b ? x0r + xa : 0.0; u = (s < q) ? ua : 0.0; }
    };
    if (__ballot(seeded)) {
      refresh(seeded);
      // RESTORATION (as in the packed sim3 kernel): while a seeded multiplier is negative the most negative slot is dropped and the iterate moved to
      // the minimiser on the remaining set; after any drop x, u are rebuilt once more from the factors
      bool restoring = seeded, did = false, again = false;
#pragma unroll 1
      for (;;) {
        const double um = rmin16((s < q) ? u : 0.0);
        bool rest = restoring && (um < 0.0);
        if (rest && ++iters > max_iter) { status = WBC_QP_MAX_ITER; rest = false; restoring = false; searching = false; }
        if (!__ballot(rest)) {
          if (!__ballot(restoring && did && !again)) break;
          const bool on = restoring && did && !again;
          refresh(on);
          if (on) again = true;
          continue;
        }
        const int l = rest ? __ffs((int)((__ballot(rest && s < q && u == um) >> rbase) & 0xFFFFull)) - 1 : 0;
        const int lcode = bpermi(a_code, rbase + (l < 0 ? 0 : l));
        drop_slot(rest, l < 0 ? 0 : l);
        const int ip = lcode & 255;
        double d = ((lcode >> 8) ? -1.0 : 1.0) * J[(ip & 15) * XLD + s];
        if (!has_b || !rest) d = 0.0;
        WSYNC();
        dv[s] = d; yv[s] = (s >= q) ? d : 0.0;
        WSYNC();
        const Zr zr = products(__ballot(rest && q > 0) != 0);
        if (rest) { x = fma(-um, zr.z, x); u = fma(um, zr.rv, u); did = true; }
      }
    }
  }

#pragma unroll 1
  for (;;) {
    // most violated inactive bound of each row
    double best = 0.0; int code = -1;
    double cand_b = 0.0;
    if (has_b && !act_b) {
      if (lb > -QP_INF) { const double sl = x - lb; if (sl < -1e-9 * fmax(1.0, fabs(lb)) && sl < best) { best = sl; code = s; cand_b = lb; } }
      if (ub < QP_INF) { const double sl = ub - x; if (sl < -1e-9 * fmax(1.0, fabs(ub)) && sl < best) { best = sl; code = s | 256; cand_b = -ub; } }
    }
    const double worst = rmin16(best);
    if (searching && !(worst < 0.0)) searching = false;               // primal feasible -> this row is optimal
#ifdef WBC_ABLATE
    if (A.dbg_stop == 308) searching = false;                         // timing cut: one violation scan, no working-set change
#endif
    if (!__ballot(searching)) break;
    const unsigned long long wm = __ballot(searching && best == worst);
    const int wl = __ffs((int)((wm >> rbase) & 0xFFFFull)) - 1;      // first lane of the row holding the worst violation
    const int wsrc = rbase + (wl < 0 ? 0 : wl);
    const int wc = bpermi(code, wsrc);
    const double b_ip = bperm(cand_b, wsrc);
    const int ip = wc & 255, ip_side = (wc >> 8) & 1;
    const double sgn = ip_side ? -1.0 : 1.0;
    double s_ip = worst, u_ip = 0.0;
    bool stepping = searching;              // row inside the partial-step loop for its bound
    int drop_l = -1;
#pragma unroll 1
    for (;;) {
      if (stepping && ++iters > max_iter) { status = WBC_QP_MAX_ITER; stepping = false; searching = false; }
      // ---- drop slot l of the rows that ask for it
      if (__ballot(stepping && drop_l >= 0)) {
        const bool dr = stepping && drop_l >= 0;
        drop_slot(dr, drop_l);
        xv[s] = x;
        WSYNC();
        if (dr) { s_ip = sgn * xv[ip & 15] - b_ip; drop_l = -1; }   // slack of the bound being added, at the current x
      }
      if (!__ballot(stepping)) break;
      // ---- d = J'n, z = J2 d2, r = T d1
      double d = sgn * J[(ip & 15) * XLD + s];
      if (!has_b || !stepping) d = 0.0;
      WSYNC();
      dv[s] = d; yv[s] = (s >= q) ? d : 0.0;
      WSYNC();
      const double zn = rsum16(s >= q ? d * d : 0.0);
      const Zr zr = products(__ballot(stepping && q > 0) != 0);
      const double z = zr.z, rv = zr.rv;
      const bool have_step = zn > 100.0 * nk * EPS2 * jf2;
      const bool cand = (s < q) && (rv > 2.2250738585072014e-308);   // (normal: frcp's estimate of a denormal is inf)
      const double ratio = cand ? u * frcp(rv) : INFINITY;
      const double t1 = rmin16(ratio);
      const unsigned long long lm = __ballot(cand && ratio == t1);
      const int l = (t1 < INFINITY) ? __ffs((int)((lm >> rbase) & 0xFFFFull)) - 1 : -1;
      const double t2 = have_step ? -s_ip * frcp(zn) : INFINITY;
      const double t = fmin(t1, t2);
      if (stepping && !(t < INFINITY)) { status = WBC_QP_INFEASIBLE; stepping = false; searching = false; }
      if (stepping) {
        if (have_step) x = fma(t, z, x);
        u = fma(-t, rv, u);
        u_ip += t;
      }
      bool add = stepping && have_step && t == t2;
      if (add && q >= XTC) { overflow = true; add = false; stepping = false; searching = false; }   // (more active bounds than T holds: the tail redoes it)
      if (__ballot(add)) {
        add_step(add, zn, zr, wc, ip, u_ip);
        if (add) stepping = false;          // this row goes back to the search
      }
      if (stepping) drop_l = l;             // blocking slot: dropped at the top of the next pass, then the step is retried
    }
  }
  if (status == WBC_QP_OPTIMAL) {
    const unsigned long long bad = __ballot(has_b && !(fabs(x) <= 1.7976931348623157e308));
    if ((bad >> rbase) & 0xFFFFull) status = WBC_QP_NUMERICAL;
  }
  // ---- x_E = -L^-T (L^-1 g_E + W~ x_K): eight row sums, one back substitution (every lane; lane e keeps entry e). The eliminated DoF's own velocity
  // bounds (never active on this controller's workloads) decide whether the reduction was valid
  double xe;
  {
    const double xk = has_b ? x : 0.0;
    double ve[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) ve[k] = rsum16(wk[k] * xk) + gtv[k];
#pragma unroll
    for (int i = 7; i >= 0; --i) {
      double t_ = ve[i];
#pragma unroll
      for (int k = i + 1; k < 8; ++k) t_ = fma(-Lk[k * 8 + i], ve[k], t_);
      ve[i] = t_ * Lk[64 + i];
    }
    double t0 = ve[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) t0 = (s == k) ? ve[k] : t0;
    xe = -t0;
  }
  if (status == WBC_QP_OPTIMAL) {
    const unsigned long long bad = __ballot(s < ne && !(fabs(xe) <= 1.7976931348623157e308));
    if ((bad >> rbase) & 0xFFFFull) status = WBC_QP_NUMERICAL;
  }
  bool flagged = false;
  {
    const bool out = (s < ne) && ((elb > -QP_INF && xe - elb < -1e-9 * fmax(1.0, fabs(elb))) || (eub < QP_INF && eub - xe < -1e-9 * fmax(1.0, fabs(eub))));
    const unsigned long long fm = __ballot(valid && status == WBC_QP_OPTIMAL && (out || overflow));
    flagged = ((fm >> rbase) & 0xFFFFull) != 0;
  }
  if (status != WBC_QP_OPTIMAL) { x = 0.0; xe = 0.0; }
  if (WARM && A.ws_out) {   // the final working set in FULL-problem indexing (word 0: bounds by DoF); an unsolved QP carries nothing; a flagged instance's is the tail's
    const int cc = a_code & 255, sd = (a_code >> 8) & 1;
    const int dA = bpermi(kv.dof, rbase + (cc & 15));
    unsigned long long w0 = 0ull;
    if (status == WBC_QP_OPTIMAL && s < q) w0 = 1ull << (32 * sd + (dA & 31));
    w0 = ror16(w0);
    if (valid && !flagged && s == 0) { A.ws_out[2 * (size_t)b] = w0; A.ws_out[2 * (size_t)b + 1] = 0ull; }
  }
  // ---- qdot by DoF through LDS, outputs
  WSYNC();
  cl[s] = 0.0; cl[16 + s] = 0.0;
  WSYNC();
  if (s < nk) cl[kv.dof & 31] = x;
  if (s < ne) cl[ev.dof & 31] = xe;
  WSYNC();
  const bool wr = valid && !flagged;
  if (wr) {
    double* qo = A.out.qdot + (size_t)b * NV;
    qo[s] = cl[s];
    if (16 + s < NV) qo[16 + s] = cl[16 + s];
    if (s == 0) {
      A.out.status[b] = status;
      if (A.out.iters) A.out.iters[b] = iters + P.x_nlock;
    }
  }
  if (A.out.q_next) {   // jointVelocitiestoConfig (Robot_Wrapper4.py:440-441)
    WSYNC();
    yv[s] = (s < 6) ? cl[s] * dt : 0.0;      // (xv shares cl's memory)
    WSYNC();
    double* qn = A.out.q_next + (size_t)b * NQ;
    if (wr) {
      const XIntegrate S{I.in, yv};
      integrate_ff(S, s, qn);
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int d = 6 + s + 16 * h2;
        if (d < nv) { const int qi = M.col_q[d]; qn[qi] = qv[qi] + cl[d] * dt; }
      }
      if (s < NQ - nq) qn[nq + s] = 0.0;
    }
  }
  // ---- the tail: instances whose reduction did not hold are redone by this wave on the general path
  const unsigned long long tailm = __ballot(valid && flagged && s == 0);
  if (tailm) {
    if (valid && flagged && s == 0 && A.defer_stat) {
      unsigned long long old = *(volatile unsigned long long*)A.defer_stat, assumed;
      do {
        assumed = old;
        const unsigned long long cnt = ((assumed >> 32) == (unsigned long long)A.tick_seq) ? (assumed & 0xFFFFFFFFull) + 1ull : 1ull;
        old = atomicCAS(A.defer_stat, assumed, ((unsigned long long)A.tick_seq << 32) | cnt);
      } while (old != assumed);
    }
#pragma unroll 1
    for (int rr = 0; rr < 4; ++rr) {
      if (!((tailm >> (16 * rr)) & 1ull)) continue;
      tail_instance<WARM, false>(&SU.G, 4 * (int)blockIdx.x + rr, models, cfgs, plans);
    }
  }
}
template __global__ void wbc_tick_boxp_kernel<false>(const KernelArgs, const DevModel* __restrict__, const WbcConfig* __restrict__, const DevPlan* __restrict__);
template __global__ void wbc_tick_boxp_kernel<true>(const KernelArgs, const DevModel* __restrict__, const WbcConfig* __restrict__, const DevPlan* __restrict__);

static int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  (void)what;
  return e == hipSuccess ? 0 : (int)e;
}

int launch_tick(const KernelArgs& a, int mode, int grid, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (mode == MODE_TICK && (a.ws_in || a.ws_out)) hipLaunchKernelGGL((wbc_tick_kernel<MODE_TICK, true>), dim3(grid), dim3(64), 0, s, a, a.models, a.cfgs, a.plans);
  else if (mode == MODE_TICK && a.presolve && a.presolve_orth == 2) hipLaunchKernelGGL((wbc_tick_kernel<MODE_TICK, false, true>), dim3(grid), dim3(64), 0, s, a, a.models, a.cfgs, a.plans);
  else if (mode == MODE_TICK) hipLaunchKernelGGL(wbc_tick_kernel<MODE_TICK>, dim3(grid), dim3(64), 0, s, a, a.models, a.cfgs, a.plans);
  else if (mode == MODE_ASSEMBLE) hipLaunchKernelGGL(wbc_tick_kernel<MODE_ASSEMBLE>, dim3(grid), dim3(64), 0, s, a, a.models, a.cfgs, a.plans);
  else hipLaunchKernelGGL(wbc_tick_kernel<MODE_FK>, dim3(grid), dim3(64), 0, s, a, a.models, a.cfgs, a.plans);
  return check_launch("tick");
}
int launch_update_packed(const UpdateArgs& a, void* stream) {
  hipLaunchKernelGGL(wbc_update_packed_kernel, dim3((a.B + 3) / 4), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  return check_launch("update_packed");
}
int launch_tick_sim3(const KernelArgs& a, int grid, void* stream) {
  if (a.ws_in || a.ws_out) hipLaunchKernelGGL(wbc_tick_sim3_kernel<true>, dim3(grid), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  else hipLaunchKernelGGL(wbc_tick_sim3_kernel<false>, dim3(grid), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  return check_launch("tick_sim3");
}
int launch_tick_deferred(const KernelArgs& a, void* stream) {
  const int grid = a.B < 2048 ? a.B : 2048;    // 8 general-path workgroups per CU: one round of the chip
  hipLaunchKernelGGL(wbc_tick_deferred_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  return check_launch("tick_deferred");
}
int launch_tick_sim3p(const KernelArgs& a, void* stream) {
  const bool warm = a.ws_in || a.ws_out, trunk = a.in.trunk_target && a.packed_trunk, qcon = a.in.q_con || a.in.posture_u;
  const dim3 grid((a.B + 3) / 4);
  if (qcon && warm) hipLaunchKernelGGL((wbc_tick_sim3p_kernel<true, false, true>), grid, dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  else if (qcon) hipLaunchKernelGGL((wbc_tick_sim3p_kernel<false, false, true>), grid, dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  else if (warm && trunk) hipLaunchKernelGGL((wbc_tick_sim3p_kernel<true, true>), grid, dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  else if (trunk) hipLaunchKernelGGL((wbc_tick_sim3p_kernel<false, true>), grid, dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  else if (warm) hipLaunchKernelGGL((wbc_tick_sim3p_kernel<true, false>), grid, dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  else hipLaunchKernelGGL((wbc_tick_sim3p_kernel<false, false>), grid, dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  return check_launch("tick_sim3p");
}
int launch_tick_orthp(const KernelArgs& a, void* stream, int ineq) {
  if (ineq && (a.ws_in || a.ws_out)) hipLaunchKernelGGL((wbc_tick_orthp_kernel<true, true>), dim3((a.B + 3) / 4), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  else if (ineq) hipLaunchKernelGGL(wbc_tick_orthp_kernel<true>, dim3((a.B + 3) / 4), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  else hipLaunchKernelGGL(wbc_tick_orthp_kernel<false>, dim3((a.B + 3) / 4), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  return check_launch("tick_orthp");
}
int orthp_lds_bytes() { return (int)(4 * sizeof(QInst)); }
int launch_tick_boxp(const KernelArgs& a, void* stream) {
  if (a.ws_in || a.ws_out) hipLaunchKernelGGL(wbc_tick_boxp_kernel<true>, dim3((a.B + 3) / 4), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  else hipLaunchKernelGGL(wbc_tick_boxp_kernel<false>, dim3((a.B + 3) / 4), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  return check_launch("tick_boxp");
}
int sim3_lds_bytes() { return (int)sizeof(SmemC); }
int sim3p_lds_bytes() { return (int)sizeof(SmemP); }
int launch_qp(const QpArgs& a, int grid, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (a.ws_in || a.ws_out) {   // hot start: the core sizes the tick problems come in (a reduced problem fits 16, the full one needs 26)
    if (a.n <= 16) hipLaunchKernelGGL((wbc_qp_kernel<16, true>), dim3(grid), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((wbc_qp_kernel<NV, true>), dim3(grid), dim3(64), 0, s, a);
  }
  else if (a.n <= 12) hipLaunchKernelGGL(wbc_qp_kernel<12>, dim3(grid), dim3(64), 0, s, a);
  else if (a.n <= 16) hipLaunchKernelGGL(wbc_qp_kernel<16>, dim3(grid), dim3(64), 0, s, a);
  else if (a.n <= 24) hipLaunchKernelGGL(wbc_qp_kernel<24>, dim3(grid), dim3(64), 0, s, a);
  else hipLaunchKernelGGL(wbc_qp_kernel<NV>, dim3(grid), dim3(64), 0, s, a);
  return check_launch("qp");
}
int launch_integrate(const IntegrateArgs& a, int grid, void* stream) {
  hipLaunchKernelGGL(wbc_integrate_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, a);
  return check_launch("integrate");
}
int launch_posture(const PostureArgs& a, int grid, void* stream) {
  hipLaunchKernelGGL(wbc_posture_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs);
  return check_launch("posture");
}
int launch_posture_par(const PostureArgs& a, int grid, void* stream, int three) {
  if (three) hipLaunchKernelGGL(wbc_posture_par3_kernel, dim3((a.B + 2) / 3), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  else hipLaunchKernelGGL(wbc_posture_par_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  return check_launch("posture_par");
}
int launch_update(const UpdateArgs& a, int grid, void* stream) {
  hipLaunchKernelGGL(wbc_update_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs);
  return check_launch("update");
}
int tick_lds_bytes() { return (int)sizeof(Smem); }

}  // namespace wbc
