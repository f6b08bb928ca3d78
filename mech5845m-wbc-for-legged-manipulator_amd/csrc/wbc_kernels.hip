// wbc_kernels.hip — gfx950 (MI355X, CDNA4) kernels of the batched whole-body-control tick.
//
// One robot instance per 64-lane wavefront, one wavefront per workgroup, persistent over the batch:
//   lane j  <-> joint j          during forward kinematics (level-synchronous over the tree depth),
//   lane k  <-> velocity DoF k   everywhere else (column k of every Jacobian, row/column k of H, J, T).
// All per-instance matrices live in LDS (row stride 26 doubles: 26 ≡ 2 mod 4 makes both the "lane = row,
// ds_read_b128 along the row" and the "lane = column, ds_read_b64 down the column" patterns bank-conflict
// free on the 64-bank LDS of CDNA4), all lane-distributed vectors in VGPRs, wave-uniform scalars in SGPRs
// via readfirstlane. HBM traffic per tick is the instance's own inputs/outputs only (~0.6-0.8 KB, coalesced).
//
// Reference semantics (file:line relative to the reference repo) are cited at each stage; the CPU restatement
// the tests compare against is oracle/wbc_oracle.c (never linked here).
#include <hip/hip_runtime.h>
#include <math.h>
#include "wbc_device.h"

namespace wbc {

constexpr int LDJ = 26;                 // LDS row stride (doubles) of Jm, T, Cm
constexpr int PMAX = WBC_MAX_P;         // 24
constexpr int MTP_MAX = 42;             // Cartesian task rows (<= 39) padded to ≡ 2 mod 4
constexpr double QP_INF = 1e20;
constexpr double EPS2 = 2.220446049250313e-16 * 2.220446049250313e-16;

struct __attribute__((aligned(16))) Smem {
  double Jm[NV * LDJ];                  // H -> L -> J = L^-T Q (n x n)
  double U[NV * LDJ + PMAX * LDJ];      // [T = R^-1 (26x26) | Cm (p x 26)]; during assembly: oMi, m*c, At
  double qv[32];
  double pf[WBC_MAX_FRAMES * 3];        // frame origins
  double dv[32], xv[32], npv[32], lv[32], dinv[32];
  double bt[48];                        // Cartesian task targets (b of qpb), uniform values
};
constexpr int OFF_T = 0, OFF_CM = NV * LDJ;
constexpr int OFF_OMI = OFF_CM;         // oMi[24][12] aliases Cm (dead before Cm is written)
constexpr int OFF_MC = 0;               // m*c per joint [32][4] aliases T (dead before T is used)
constexpr int OFF_AT = 0;               // At[26][mtp] aliases T|Cm (dead before either is written)

#define WSYNC() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")

// Diagnostic build (-DWBC_PROFILE): s_memtime stamps at phase boundaries, summed per phase into KernelArgs.prof.
// Never compiled into the shipped library; its run time is not quoted (the stamps serialise the phases).
#ifdef WBC_PROFILE
#define STAMP(ts, i) do { __builtin_amdgcn_sched_barrier(0); (ts)[i] = (unsigned long long)clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(ts, i) do { } while (0)
#endif
enum { T_START = 0, T_FK = 1, T_ASM = 2, T_CHOL = 3, T_INV = 4, T_EQ = 5, T_INEQ = 6, T_END = 7, T_N = 8 };

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
// reductions over lanes 0..31 (the 26 DoF lanes live there); result is wave-uniform
__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return rfl(v);
}
__device__ __forceinline__ double wmin(double v) {
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) v = fmin(v, __shfl_xor(v, m));
  return rfl(v);
}
__device__ __forceinline__ int ctz64(unsigned long long m) { return __ffsll((long long)m) - 1; }

struct double2a { double x, y; } __attribute__((aligned(16)));
__device__ __forceinline__ double2a lds2(const double* p) { return *reinterpret_cast<const double2a*>(p); }


__device__ __forceinline__ int li_clamp(int lane) { return lane < NV ? lane : NV - 1; }

__device__ __forceinline__ void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
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

// Free-flyer part of pin.integrate (Robot_Wrapper4.py:441): M+ = M exp6(v), v = S.xv[0..5] (body twist * dt).
// R0 (row-major) / p0 = current base placement, quaternion continuity + first-order renormalisation as in
// pinocchio's SpecialEuclideanOperationTpl<3>::integrate_impl. Uniform arithmetic; lanes 0..6 store.
__device__ __forceinline__ void integrate_ff(const Smem& S, const int lane, const double* R0, const double* p0, double* qn) {
  const double vl[3] = {S.xv[0], S.xv[1], S.xv[2]}, w[3] = {S.xv[3], S.xv[4], S.xv[5]};
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], t = sqrt(t2);
  double a, bq, c;
  if (t < 1e-4) { a = 1 - t2 / 6; bq = 0.5 - t2 / 24; c = 1.0 / 6 - t2 / 120; }
  else { double sn, cs; sincos(t, &sn, &cs); a = sn / t; bq = (1 - cs) / t2; c = (1 - a) / t2; }
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
  double qq[4];
  const double tr = R1[0] + R1[4] + R1[8];
  if (tr > 0) {
    double s = sqrt(tr + 1.0);
    qq[3] = 0.5 * s; s = 0.5 / s;
    qq[0] = (R1[7] - R1[5]) * s; qq[1] = (R1[2] - R1[6]) * s; qq[2] = (R1[3] - R1[1]) * s;
  } else if (R1[0] >= R1[4] && R1[0] >= R1[8]) {   // i = 0 (Eigen: i=0; if m11>m00 i=1; if m22>m_ii i=2)
    double s = sqrt(R1[0] - R1[4] - R1[8] + 1.0);
    qq[0] = 0.5 * s; s = 0.5 / s;
    qq[3] = (R1[7] - R1[5]) * s; qq[1] = (R1[3] + R1[1]) * s; qq[2] = (R1[6] + R1[2]) * s;
  } else if (R1[4] > R1[0] && R1[4] >= R1[8]) {    // i = 1
    double s = sqrt(R1[4] - R1[8] - R1[0] + 1.0);
    qq[1] = 0.5 * s; s = 0.5 / s;
    qq[3] = (R1[2] - R1[6]) * s; qq[2] = (R1[7] + R1[5]) * s; qq[0] = (R1[1] + R1[3]) * s;
  } else {                                         // i = 2
    double s = sqrt(R1[8] - R1[0] - R1[4] + 1.0);
    qq[2] = 0.5 * s; s = 0.5 / s;
    qq[3] = (R1[3] - R1[1]) * s; qq[0] = (R1[2] + R1[6]) * s; qq[1] = (R1[5] + R1[7]) * s;
  }
  if (qq[0] * S.qv[3] + qq[1] * S.qv[4] + qq[2] * S.qv[5] + qq[3] * S.qv[6] < 0) { qq[0] = -qq[0]; qq[1] = -qq[1]; qq[2] = -qq[2]; qq[3] = -qq[3]; }
  const double f = (3 - (qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3])) / 2;
  double outv = 0.0;
  if (lane == 0) outv = pn[0];
  if (lane == 1) outv = pn[1];
  if (lane == 2) outv = pn[2];
  if (lane == 3) outv = qq[0] * f;
  if (lane == 4) outv = qq[1] * f;
  if (lane == 5) outv = qq[2] * f;
  if (lane == 6) outv = qq[3] * f;
  if (lane < 7) qn[lane] = outv;
}

// ------------------------------------------------------------------------------------------------
// QP core: Goldfarb–Idnani dual active set, wavefront form (algebra: tests/gi_variant.py).
//   in : h[26]  row `lane` of H (registers), g, lb, ub per lane; Cm (p x 26) in LDS, clb/cub per lane (row = lane)
//   out: x per lane; returns status; iters
// replaces qpOASES init/hotstart as called at QP_Wrapper.py:45-48, 70 (unique minimiser since H > 0).
// ------------------------------------------------------------------------------------------------
struct QpResult { double x; int status; int iters; };

__device__ __forceinline__ QpResult qp_core(Smem& S, double (&h)[NV], const double g, const double lb, const double ub,
                                         const double clb, const double cub, const int n, const int p, const int lane,
                                         unsigned long long* ts) {
  const int li = lane < NV ? lane : NV - 1;           // clamped lane for LDS reads
  double* Jm = S.Jm;
  double* T = S.U + OFF_T;
  const double* Cm = S.U + OFF_CM;
  QpResult res;
  res.status = WBC_QP_OPTIMAL;
  res.iters = 0;

  // ---- Cholesky H = L L' (right-looking, lane i owns row i in registers; column j broadcast through LDS)
  // PIN(): zero-instruction use/def that stops the compiler from sinking a step's arithmetic below later steps
  // (it otherwise keeps every broadcast column alive at once and spills hundreds of registers).
#define PIN(v) asm volatile("" : "+v"(v))
  double pmin = 1.0;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    if (lane < NV) S.lv[lane] = h[j];
    WSYNC();
    const double pj = S.lv[j];
    pmin = fmin(pmin, pj);
    const double rinv = rsqrt(pj);
    if (lane == 0) S.dinv[j] = rinv;
    const double lij = h[j] * rinv;
    h[j] = lij;
#pragma unroll
    for (int k = j + 1; k < NV; ++k) h[k] = fma(-lij, S.lv[k] * rinv, h[k]);
    WSYNC();
    PIN(pmin);
#pragma unroll
    for (int k = j + 1; k < NV; ++k) PIN(h[k]);
  }
  STAMP(ts, T_CHOL);
  if (!(pmin > 0.0)) { res.status = WBC_QP_NUMERICAL; res.x = 0.0; return res; }
  // L rows -> LDS (zero above the diagonal)
  if (lane < NV) {
#pragma unroll
    for (int k = 0; k < NV; ++k) Jm[lane * LDJ + k] = (k <= lane) ? h[k] : 0.0;
  }
  WSYNC();
  // ---- J = L^-T: lane c solves L y = e_c; y = column c of L^-1 = row c of J
  double y[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double s = (i == lane) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < i; ++k) s = fma(-Jm[i * LDJ + k], y[k], s);
    y[i] = s * S.dinv[i];
    if ((i & 3) == 3) asm volatile("" : "+v"(y[i]) : : "memory");   // bound how far row loads are hoisted
  }
  WSYNC();
  double sq = 0.0;
  if (lane < NV) {
#pragma unroll
    for (int k = 0; k < NV; ++k) { Jm[lane * LDJ + k] = y[k]; sq = fma(y[k], y[k], sq); }
  }
  const double jf2 = wsum(lane < NV ? sq : 0.0);
  // ---- x = -J J' g
  if (lane < 32) S.npv[lane] = (lane < NV) ? g : 0.0;
  WSYNC();
  double dg = 0.0;
#pragma unroll
  for (int i = 0; i < NV; ++i) dg = fma(Jm[i * LDJ + li], S.npv[i], dg);
  if (lane < 32) S.dv[lane] = (lane < NV) ? dg : 0.0;
  WSYNC();
  double x = 0.0;
#pragma unroll
  for (int k = 0; k < NV; ++k) x = fma(-y[k], S.dv[k], x);
  if (lane >= n) x = 0.0;
  STAMP(ts, T_INV);
  // ---- T = 0
  for (int k = lane; k < NV * LDJ; k += 64) T[k] = 0.0;

  // ---- active-set state
  const bool has_b = lane < n, has_r = lane < p;
  const bool eq_b = has_b && (lb == ub) && (fabs(lb) < QP_INF);
  const bool eq_r = has_r && (clb == cub) && (fabs(clb) < QP_INF);
  unsigned long long eqm_b = __ballot(eq_b), eqm_r = __ballot(eq_r);
  bool act_b = false, act_r = false;      // bound `lane` / row `lane` in the working set
  double u = 0.0;                         // multiplier of working-set slot `lane`
  int a_code = 0;                         // slot `lane`: constraint id | side << 8 | eq << 9
  int q = 0, iters = 0;
#ifdef WBC_PROFILE
  bool eq_done = false;
#endif
  const int max_iter = 10 * (n + p) + 20;
  double cn2 = 0.0;                       // |C_r|^2 for row = lane
  if (has_r) {
#pragma unroll
    for (int k = 0; k < NV; k += 2) { const double2a c2 = lds2(Cm + lane * LDJ + k); cn2 = fma(c2.x, c2.x, fma(c2.y, c2.y, cn2)); }
  }
  WSYNC();

  for (;;) {
    // ---------------- choose the constraint to add
    int ip, ip_side = 0, ip_eq = 0;
    double s_ip, b_ip;
    if (eqm_b) {                                        // equalities in index order: bounds first
      ip = ctz64(eqm_b); eqm_b &= eqm_b - 1; ip_eq = 1;
      b_ip = rdl(lb, ip);
      s_ip = rdl(x, ip) - b_ip;
    } else if (eqm_r) {
      const int r = ctz64(eqm_r); eqm_r &= eqm_r - 1; ip_eq = 1; ip = n + r;
      b_ip = rdl(clb, r);
      s_ip = wsum(lane < n ? Cm[r * LDJ + li] * x : 0.0) - b_ip;
    } else {                                            // most violated inactive inequality
#ifdef WBC_PROFILE
      if (!eq_done) { eq_done = true; STAMP(ts, T_EQ); }
#endif
      if (lane < 32) S.xv[lane] = (lane < n) ? x : 0.0;
      WSYNC();
      double best = 0.0; int code = -1;
      if (has_b && !act_b && !eq_b) {
        if (lb > -QP_INF) { const double s = x - lb; if (s < -1e-9 * fmax(1.0, fabs(lb)) && s < best) { best = s; code = lane; } }
        if (ub < QP_INF) { const double s = ub - x; if (s < -1e-9 * fmax(1.0, fabs(ub)) && s < best) { best = s; code = lane | 256; } }
      }
      if (p > 0) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < NV; k += 2) {
          const double2a c2 = lds2(Cm + (has_r ? lane : 0) * LDJ + k); const double2a x2 = lds2(S.xv + k);
          v = fma(c2.x, x2.x, fma(c2.y, x2.y, v));
        }
        if (has_r && !act_r && !eq_r) {
          if (clb > -QP_INF) { const double s = v - clb; if (s < -1e-9 * fmax(1.0, fabs(clb)) && s < best) { best = s; code = n + lane; } }
          if (cub < QP_INF) { const double s = cub - v; if (s < -1e-9 * fmax(1.0, fabs(cub)) && s < best) { best = s; code = (n + lane) | 256; } }
        }
      }
      const double worst = wmin(lane < 32 ? best : 0.0);
      if (!(worst < 0.0)) break;                        // primal feasible -> optimal
      const int wl = ctz64(__ballot(lane < 32 && best == worst));
      const int wc = rdli(code, wl);
      ip = wc & 255; ip_side = (wc >> 8) & 1;
      s_ip = worst;
      const double bl = (ip < n) ? rdl(ip_side ? -ub : lb, ip) : rdl(ip_side ? -cub : clb, ip - n);
      b_ip = bl;
    }
    const double sgn = ip_side ? -1.0 : 1.0;
    const bool is_row = ip >= n;
    const int rr = is_row ? ip - n : 0;
    const double np2 = is_row ? rdl(cn2, rr) : 1.0;
    double u_ip = 0.0;
    bool added_or_skipped = false;

    while (!added_or_skipped) {
      if (++iters > max_iter) { res.status = WBC_QP_MAX_ITER; goto done; }
      // d = J' np  (lane k: column k of J)
      double d = 0.0;
      if (is_row) {
#pragma unroll
        for (int i = 0; i < NV; i += 2) {
          const double2a c2 = lds2(Cm + rr * LDJ + i);
          d = fma(Jm[i * LDJ + li], c2.x, fma(Jm[(i + 1) * LDJ + li], c2.y, d));
        }
        d *= sgn;
      } else {
        d = sgn * Jm[ip * LDJ + li];
      }
      if (lane >= n) d = 0.0;
      if (lane < 32) S.dv[lane] = d;
      WSYNC();
      const double zn = wsum((lane >= q && lane < n) ? d * d : 0.0);
      // z = J2 d2 (lane i: row i of J), r = T d1 (lane i < q: row i of T)
      double z = 0.0, r = 0.0;
      {
        const int k0 = q & ~1;
        for (int k = k0; k < NV; k += 2) {
          const double2a j2 = lds2(Jm + li * LDJ + k); const double2a d2 = lds2(S.dv + k);
          z = fma(j2.x, (k >= q) ? d2.x : 0.0, fma(j2.y, d2.y, z));
        }
        for (int k = 0; k < q; k += 2) {
          const double2a t2 = lds2(T + li * LDJ + k); const double2a d2 = lds2(S.dv + k);
          r = fma(t2.x, d2.x, fma(t2.y, (k + 1 < q) ? d2.y : 0.0, r));
        }
        if (lane >= q) r = 0.0;
        if (lane >= n) z = 0.0;
      }
      const bool have_step = zn > 100.0 * n * EPS2 * jf2 * np2;
      // dual step length t1 = min u_k / r_k over inequality slots with r_k > 0
      const bool cand = (lane < q) && !((a_code >> 9) & 1) && (r > 0.0);
      const double ratio = cand ? u / r : INFINITY;
      const double t1 = wmin(lane < 32 ? ratio : INFINITY);
      const int l = (t1 < INFINITY) ? ctz64(__ballot(cand && ratio == t1)) : -1;
      const double t2 = have_step ? -s_ip / zn : INFINITY;
      if (ip_eq && !have_step) {                        // dependent equality
        if (fabs(s_ip) <= 1e-9 * fmax(1.0, fabs(b_ip))) { added_or_skipped = true; break; }
        res.status = WBC_QP_INFEASIBLE; goto done;
      }
      const double t = ip_eq ? t2 : fmin(t1, t2);
      if (!(t < INFINITY)) { res.status = WBC_QP_INFEASIBLE; goto done; }
      if (have_step) x = fma(t, z, x);
      u = fma(-t, r, u);
      u_ip += t;
      if (have_step && t == t2) {
        // ---- add: Householder P with P d2 = delta e1; J2 <- J2 P; T gets column (-r/delta, 1/delta)
        const double dq = rdl(d, q);
        const double sz = sqrt(zn);
        const double delta = (dq >= 0.0) ? -sz : sz;
        const double vv = 2.0 * (zn - delta * dq);
        if (vv > 0.0) {
          const double beta = 2.0 / vv;
          const double w = (z - delta * Jm[li * LDJ + q]) * beta;
          if (lane < n) {
            for (int k = q; k < n; ++k) {
              const double vk = S.dv[k] - ((k == q) ? delta : 0.0);
              Jm[lane * LDJ + k] = fma(-w, vk, Jm[lane * LDJ + k]);
            }
          }
        }
        const double idel = 1.0 / delta;
        if (lane < q) T[lane * LDJ + q] = -r * idel;
        if (lane == q) { T[lane * LDJ + q] = idel; u = u_ip; a_code = ip | (ip_side << 8) | (ip_eq << 9); }
        if (is_row) { if (lane == rr) act_r = true; } else { if (lane == ip) act_b = true; }
        ++q;
        added_or_skipped = true;
        WSYNC();
      } else {
        // ---- drop slot l: Givens sequence read off the removed row of T, applied to columns of T and J
        const int lc = rdli(a_code, l) & 255;
        if (lc >= n) { if (lane == lc - n) act_r = false; } else { if (lane == lc) act_b = false; }
        {
          const double un = __shfl_down(u, 1); const int an = __shfl_down(a_code, 1);
          if (lane >= l && lane < q - 1) { u = un; a_code = an; }
          if (lane == q - 1) { u = 0.0; a_code = 0; }
        }
        const int srow = (li >= l) ? ((li + 1 < NV) ? li + 1 : li) : li;   // old row feeding new row `lane`
        double tx = T[srow * LDJ + l];
        double jx = Jm[li * LDJ + l];
        double hrun = T[l * LDJ + l];
        for (int k = l; k < q - 1; ++k) {
          const double tb = T[l * LDJ + k + 1];
          const double nrm2 = fma(hrun, hrun, tb * tb);
          double c_ = 1.0, s_ = 0.0, rho = 0.0;
          if (nrm2 > 0.0) { const double ri = rsqrt(nrm2); c_ = tb * ri; s_ = -hrun * ri; rho = nrm2 * ri; }
          hrun = rho;
          const double ty = T[srow * LDJ + k + 1];
          const double jy = Jm[li * LDJ + k + 1];
          WSYNC();
          if (lane < q - 1) T[lane * LDJ + k] = fma(c_, tx, s_ * ty);
          if (lane < n) Jm[lane * LDJ + k] = fma(c_, jx, s_ * jy);
          tx = fma(-s_, tx, c_ * ty);
          jx = fma(-s_, jx, c_ * jy);
        }
        WSYNC();
        if (lane < q) T[lane * LDJ + q - 1] = 0.0;      // dropped last column, and the vacated last row
        if (lane < q) T[(q - 1) * LDJ + lane] = 0.0;
        if (lane < n) Jm[lane * LDJ + q - 1] = jx;
        --q;
        WSYNC();
        // constraint ip's slack at the new x
        const double v = is_row ? wsum(lane < n ? Cm[rr * LDJ + li] * x : 0.0) : rdl(x, ip);
        s_ip = sgn * v - b_ip;
      }
    }
  }
done:
  STAMP(ts, T_INEQ);
  res.x = x;
  res.iters = iters;
  return res;
}

// ------------------------------------------------------------------------------------------------
// J'J on the fp64 matrix cores: H = A'A with v_mfma_f64_16x16x4_f64 (QP_Wrapper.py:17: np.dot(A.T, A)).
// A is m x n (n <= 26, padded to 32 = 2 x 16 columns); k-step s contracts task rows 4s..4s+3.
// Operand maps (cdna_hip_programming.md §3): lane l feeds A_op[i = l&15][k = l>>4] and B_op[k = l>>4][j = l&15], so
// for tile (I, J) both operands are one double per lane: A[4s + (l>>4)][16 I/J + (l&15)]. D: lane l, reg r holds
// D[(l>>4) + 4r][l&15]. Tiles 00, 01, 11 are computed (10 = 01'). `load(r, c)` returns A[r][c] (0 outside);
// column 26 may carry b so that -A'b falls out of the same MFMAs (written to S.npv as +A'b).
// The tiles go to S.Jm (row-major, stride LDJ); every lane then reads its row of H into h[].
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <class LoadA>
__device__ __forceinline__ void jtj_mfma(Smem& S, const int lane, const int m, LoadA load, double (&h)[NV]) {
  v4f64 acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
  const int kq = lane >> 4, c0 = lane & 15;
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
    S.Jm[row * LDJ + col] = acc00[r];
    if (16 + col < NV) { S.Jm[row * LDJ + 16 + col] = acc01[r]; S.Jm[(16 + col) * LDJ + row] = acc01[r]; }
    if (16 + row < NV && 16 + col < NV) S.Jm[(16 + row) * LDJ + 16 + col] = acc11[r];
    if (16 + col == NV) { S.npv[row] = acc01[r]; if (16 + row < NV) S.npv[16 + row] = acc11[r]; }
  }
  WSYNC();
  const int li = li_clamp(lane);
#pragma unroll
  for (int k = 0; k < NV; k += 2) { const double2a v = lds2(S.Jm + li * LDJ + k); h[k] = v.x; h[k + 1] = v.y; }
  WSYNC();
}

// ------------------------------------------------------------------------------------------------
// H += A_t' A_t for one block of `nr` task rows [row0, row0+nr) over the DoF set `mask`.
// lane k holds its own column a[] of the block; At[i][row] (LDS) supplies the other columns, uniform address.
template <int NR>
__device__ __forceinline__ void jtj_block(const double* At, int mtp, int row0, unsigned mask, const double (&a)[NR],
                                          double (&h)[NV]) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if ((mask >> i) & 1u) {
      double s = h[i];
#pragma unroll
      for (int r = 0; r < NR; ++r) s = fma(At[i * mtp + row0 + r], a[r], s);
      h[i] = s;
    }
  }
}


// scipy Rotation.from_matrix(M).as_quat() branch logic (Robot_Wrapper4.py:964-965); M row-major.
// Written out per branch: a dynamically indexed M would be demoted to scratch memory.
__device__ inline void R_to_quat(const double* M, double* q) {
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
__device__ inline void quat_mul(const double* a, const double* b, double* r) {
  r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  r[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  r[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}

// ------------------------------------------------------------------------------------------------
// One instance: FK -> Jacobians -> task stack -> H, g, C, bounds [-> QP -> qdot -> q_next]
// ------------------------------------------------------------------------------------------------
template <int MODE>
__device__ void process_instance(Smem& S, const KernelArgs& A, const int b, const int lane) {
  const int mid = A.in.model_id ? A.in.model_id[b] : 0;
  const DevModel& M = A.models[mid];
  const WbcConfig& cfg = A.cfgs[mid];
  const int nv = M.nv, nq = M.nq, nj = M.njoints;
  const double dt = A.dt;
  double* oMi = S.U + OFF_OMI;          // [joint][12]: R column-major (3 columns), then p
  unsigned long long ts[T_N];
  (void)ts;
  STAMP(ts, T_START);
  const int ll = lane & 31;             // index into the 32-entry per-lane model tables

  // ---- P0: q (coalesced), updateState's config (Robot_Wrapper4.py:389-402)
  const double* qg = A.in.q + (size_t)b * NQ;
  if (lane < 32) S.qv[lane] = (lane < nq) ? qg[lane] : 0.0;
  WSYNC();

  // ---- P1: forward kinematics, pin.forwardKinematics (Robot_Wrapper4.py:400)
  // root free-flyer: R from the quaternion exactly as Eigen's toRotationMatrix, p = xyz (every lane, uniform)
  if (lane == 1) {
    double Rt[9];
    quat_to_R(S.qv + 3, Rt);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 3; ++r) oMi[12 + 3 * c + r] = Rt[3 * r + c];
    oMi[12 + 9] = S.qv[0]; oMi[12 + 10] = S.qv[1]; oMi[12 + 11] = S.qv[2];
  }
  {
    const int jt = M.jtype[ll];
    const bool is_joint = lane >= 2 && lane < nj;
    const bool rev = jt >= WBC_JT_RX && jt <= WBC_JT_RZ;
    const double th = is_joint ? S.qv[M.idx_q[ll]] : 0.0;
    double sn = 0.0, cs = 1.0;
    if (rev) sincos(th, &sn, &cs);
    const double pris = (is_joint && !rev) ? th : 0.0;
    const int a0 = 3 * M.ax0[ll], a1 = 3 * M.ax1[ll], a2 = 3 * M.ax2[ll];
    const double t0 = M.tp[ll][0], t1 = M.tp[ll][1], t2 = M.tp[ll][2];
    const int par = is_joint ? M.parent[ll] : 1, dep = is_joint ? M.depth[ll] : 0;
    WSYNC();
    for (int lvl = 2; lvl <= M.maxdepth; ++lvl) {
      if (dep == lvl) {
        const double* Pp = oMi + 12 * par;
        double Av[3], Bv[3], Cv[3], P[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) { Av[r] = Pp[a0 + r]; Bv[r] = Pp[a1 + r]; Cv[r] = Pp[a2 + r]; P[r] = Pp[9 + r]; }
        double* Po = oMi + 12 * lane;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          Po[a0 + r] = Av[r];
          Po[a1 + r] = cs * Bv[r] + sn * Cv[r];
          Po[a2 + r] = cs * Cv[r] - sn * Bv[r];
          Po[9 + r] = P[r] + Av[r] * (t0 + pris) + Bv[r] * t1 + Cv[r] * t2;
        }
      }
      WSYNC();
    }
  }
  // ---- P2: frame origins, pin.updateFramePlacements (Robot_Wrapper4.py:405); frames carry no rotation offset
  if (lane < M.nframes) {
    const double* Pj = oMi + 12 * M.frame_joint[lane];
    const double f0 = M.frame_p[lane][0], f1 = M.frame_p[lane][1], f2 = M.frame_p[lane][2];
#pragma unroll
    for (int r = 0; r < 3; ++r) S.pf[3 * lane + r] = Pj[9 + r] + Pj[r] * f0 + Pj[3 + r] * f1 + Pj[6 + r] * f2;
  }
  const bool need_com = cfg.task_com || cfg.con_com || (MODE == MODE_FK && (A.fk.com || A.fk.Jcom));
  if (need_com) {   // m_j * c_j (world) per joint, pin.jacobianCenterOfMass's subtree pass (Robot_Wrapper4.py:670)
    double* mc = S.U + OFF_MC;
    if (lane >= 1 && lane < nj) {
      const double* Pj = oMi + 12 * lane;
      const double m = M.mass[ll], c0 = M.com[ll][0], c1 = M.com[ll][1], c2 = M.com[ll][2];
#pragma unroll
      for (int r = 0; r < 3; ++r) mc[4 * lane + r] = m * (Pj[9 + r] + Pj[r] * c0 + Pj[3 + r] * c1 + Pj[6 + r] * c2);
      mc[4 * lane + 3] = m;
    }
  }
  WSYNC();
  // ---- P3: column k of data.J, pin.computeJointJacobians (Robot_Wrapper4.py:403), WORLD frame
  double lin[3] = {0, 0, 0}, ang[3] = {0, 0, 0};
  const int cj = M.col_joint[ll];
  if (lane < nv) {
    const double* Pj = oMi + 12 * cj;
    const int la = M.col_lin[ll], aa = M.col_ang[ll];
    double pj[3] = {Pj[9], Pj[10], Pj[11]};
    if (aa >= 0) { ang[0] = Pj[3 * aa]; ang[1] = Pj[3 * aa + 1]; ang[2] = Pj[3 * aa + 2]; cross3(pj, ang, lin); }
    if (la >= 0) { lin[0] = Pj[3 * la]; lin[1] = Pj[3 * la + 1]; lin[2] = Pj[3 * la + 2]; }
  }
  double com[3] = {0, 0, 0}, jc[3] = {0, 0, 0};   // whole-body CoM (uniform) and column k of Jcom
  if (need_com) {
    const double* mc = S.U + OFF_MC;
    const unsigned sub = (lane < nv) ? M.col_subtree[ll] : 0u;
    double ms = 0, s0 = 0, s1 = 0, s2 = 0;
    for (int j = 1; j < nj; ++j) {
      const double f = ((sub >> j) & 1u) ? 1.0 : 0.0;
      s0 = fma(f, mc[4 * j], s0); s1 = fma(f, mc[4 * j + 1], s1); s2 = fma(f, mc[4 * j + 2], s2); ms = fma(f, mc[4 * j + 3], ms);
    }
    const double Mt = rdl(ms, 0);
    com[0] = rdl(s0, 0) / Mt; com[1] = rdl(s1, 0) / Mt; com[2] = rdl(s2, 0) / Mt;
    if (lane < nv && ms > 0.0) {
      const double cs_[3] = {s0 / ms, s1 / ms, s2 / ms};
      double wxc[3];
      cross3(ang, cs_, wxc);
      const double f = ms / Mt;
#pragma unroll
      for (int r = 0; r < 3; ++r) jc[r] = f * (lin[r] + wxc[r]);
    }
  }
  // trunk frame (imu): rotation of its supporting joint, uniform read
  double Rtr[9], ptr[3];
  {
    const double* Pj = oMi + 12 * M.frame_joint[WBC_FR_TRUNK];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 3; ++r) Rtr[3 * r + c] = Pj[3 * c + r];
    ptr[0] = S.pf[3 * WBC_FR_TRUNK]; ptr[1] = S.pf[3 * WBC_FR_TRUNK + 1]; ptr[2] = S.pf[3 * WBC_FR_TRUNK + 2];
  }

  if (MODE == MODE_FK) {
    // outputs of updateState: oMi / oMf (row-major R then p), data.J, com, Jcom
    if (A.fk.oMi && lane < nj) {
      double* o = A.fk.oMi + ((size_t)b * A.models[0].njoints + lane) * 12;   // strides of model 0
      if (lane == 0) { for (int i = 0; i < 12; ++i) o[i] = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0; }
      else {
        const double* Pj = oMi + 12 * lane;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o[3 * r + c] = Pj[3 * c + r];
        o[9] = Pj[9]; o[10] = Pj[10]; o[11] = Pj[11];
      }
    }
    if (A.fk.oMf && lane < M.nframes) {
      double* o = A.fk.oMf + ((size_t)b * A.models[0].nframes + lane) * 12;
      const double* Pj = oMi + 12 * M.frame_joint[lane];
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
  WSYNC();   // every lane is done reading oMi / mc: the region is reused for At from here on
  double h[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) h[i] = 0.0;
  double g = 0.0;
  double* At = S.U + OFF_AT;
  const int mtp = (A.mcart + 3) / 4 * 4 + 2;     // ≡ 2 mod 4
  int row = 0;
  const bool out_A = (MODE == MODE_ASSEMBLE) && A.qp.A != nullptr;
  // pass 1: every lane writes its column of every Cartesian block to At and accumulates g
  for (int e = 0; e < WBC_NEE; ++e) {
    if (!cfg.task_ee[e]) continue;
    const bool sup = (lane < nv) && ((M.frame_support[WBC_FR_EE0 + e] >> lane) & 1u);
    const double pfe[3] = {S.pf[3 * e], S.pf[3 * e + 1], S.pf[3 * e + 2]};
    double a[6];
    {  // endEffectorA2 (Robot_Wrapper4.py:474-484): LOCAL_WORLD_ALIGNED: lin + ang x p_f
      double wxp[3];
      cross3(ang, pfe, wxp);
      const double w = cfg.ee_w[e];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        a[r] = sup ? cfg.ee_W[e][r] * ((lin[r] + wxp[r]) * w) : 0.0;
        a[3 + r] = sup ? cfg.ee_W[e][3 + r] * (ang[r] * w) : 0.0;
      }
    }
    // calcTargetVelEE3 (Robot_Wrapper4.py:1052-1157) — uniform arithmetic
    const double* xt = A.in.ee_target + ((size_t)b * WBC_NEE + e) * 3;
    const double* xp = A.in.prev_ee_target + ((size_t)b * WBC_NEE + e) * 3;
    double vel[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 3; ++i) vel[i] = (xt[i] - xp[i]) / dt + cfg.ee_gain[e][i] * ((xt[i] - pfe[i]) / dt);
    if (A.in.ee_ref_rot && A.in.ee_prev_rot) {   // omega = vee(((R* - R*_prev)/dt) R*^T)  (:1125-1128, 1133)
      const double* Rs = A.in.ee_ref_rot + ((size_t)b * WBC_NEE + e) * 9;
      const double* Rp = A.in.ee_prev_rot + ((size_t)b * WBC_NEE + e) * 9;
      double D[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) D[i] = (Rs[i] - Rp[i]) / dt;
      vel[3] = D[6] * Rs[3] + D[7] * Rs[4] + D[8] * Rs[5];   // S[2][1]
      vel[4] = D[0] * Rs[6] + D[1] * Rs[7] + D[2] * Rs[8];   // S[0][2]
      vel[5] = D[3] * Rs[0] + D[4] * Rs[1] + D[5] * Rs[2];   // S[1][0]
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double br = vel[r] * cfg.ee_w[e];                  // EndEffectorB2 (:907-910)
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
    const double* xt = A.in.trunk_target + (size_t)b * 3;
    const double* xp = A.in.prev_trunk_target + (size_t)b * 3;
    double vel[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) vel[i] = (xt[i] - xp[i]) / dt + cfg.trunk_gain[i] * ((xt[i] - ptr[i]) / dt);
    double fq[4], rq[4], Rs[9];
    R_to_quat(Rtr, fq);
    const double* er = A.in.trunk_ref_euler + (size_t)b * 3;
    {
      double sa, ca, sb, cb, sc, cc;
      sincos(er[0], &sa, &ca); sincos(er[1], &sb, &cb); sincos(er[2], &sc, &cc);
      Rs[0] = cc * cb; Rs[1] = cc * sb * sa - sc * ca; Rs[2] = cc * sb * ca + sc * sa;
      Rs[3] = sc * cb; Rs[4] = sc * sb * sa + cc * ca; Rs[5] = sc * sb * ca - cc * sa;
      Rs[6] = -sb;     Rs[7] = cb * sa;                Rs[8] = cb * ca;
      double s2, c2;
      sincos(er[0] / 2, &s2, &c2); const double qx[4] = {s2, 0, 0, c2};
      sincos(er[1] / 2, &s2, &c2); const double qy[4] = {0, s2, 0, c2};
      sincos(er[2] / 2, &s2, &c2); const double qz[4] = {0, 0, s2, c2};
      double tq[4];
      quat_mul(qy, qx, tq);
      quat_mul(qz, tq, rq);
    }
    const double qe0 = fq[3] * rq[0] - fq[0] * rq[3] + fq[1] * rq[2] - fq[2] * rq[1];   // :974
    const double qe1 = fq[3] * rq[1] - fq[1] * rq[3] - fq[0] * rq[2] + fq[2] * rq[0];   // :975
    const double qe2 = fq[3] * rq[2] - fq[3] * rq[2] + fq[0] * rq[1] - fq[1] * rq[0];   // :976 (sic)
    const double* Ro = A.in.trunk_prev_rot + (size_t)b * 9;
    double D[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) D[i] = (Rs[i] - Ro[i]) / dt;
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
    const double* ct = A.in.com_target + (size_t)b * 3;
    const double* cv = A.in.com_target_vel + (size_t)b * 3;
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
  // pass 2: H[lane][i] += sum_r At[i][r] At[lane][r]
  if (A.jtj_mfma) {
    // dense contraction on the fp64 matrix cores (the operand comes straight from the At image in LDS)
    const int mc = A.mcart;
    jtj_mfma(S, lane, mc, [&](int r, int c) -> double { return (r < mc && c < NV) ? At[c * mtp + r] : 0.0; }, h);
  } else {
    // vector units, block by block over each block's DoF support (skips the structural zeros of the Jacobians)
    int r0 = 0;
    for (int e = 0; e < WBC_NEE; ++e) {
      if (!cfg.task_ee[e]) continue;
      double a[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) a[r] = At[li_clamp(lane) * mtp + r0 + r];
      jtj_block<6>(At, mtp, r0, M.frame_support[WBC_FR_EE0 + e], a, h);
      r0 += 6;
    }
    if (cfg.task_trunk) {
      double a[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) a[r] = At[li_clamp(lane) * mtp + r0 + r];
      jtj_block<6>(At, mtp, r0, M.frame_support[WBC_FR_TRUNK], a, h);
      r0 += 6;
    }
    if (cfg.task_com) {
      double a[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) a[r] = At[li_clamp(lane) * mtp + r0 + r];
      jtj_block<3>(At, mtp, r0, (1u << nv) - 1u, a, h);
      r0 += 3;
    }
  }
  // posture rows: qpJointA (Robot_Wrapper4.py:1199-1206), qpJointb (:1209-1268)
  double dpost = 0.0, upost = 0.0;
  if (cfg.task_joint) {
    dpost = (1.0 / nv) * cfg.joint_w;
    if (cfg.task_joint == WBC_JOINT_PREV && lane < nv) upost = S.qv[lane < 6 ? lane : lane + 1];   // np.delete(q, 6)
    const double bj = (1.0 / nv) * upost * cfg.joint_w;
    if (lane < nv) g = fma(-dpost, bj, g);
    upost = bj;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (i == lane) h[i] += (lane < nv) ? dpost * dpost : 1.0;   // padded DoF: H_dd = 1
  }
  if (lane >= nv) g = 0.0;

  if (MODE == MODE_ASSEMBLE) {
    const int m = A.mrows;
    if (out_A && lane < NV) {
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
#pragma unroll
      for (int k = 0; k < NV; ++k) o[k] = h[k];
    }
    if (A.qp.g && lane < NV) A.qp.g[(size_t)b * NV + lane] = g;
  }
  WSYNC();   // At is dead: Cm may be written

  // ---- P6: constraints in order CoM, Trunk, FR, FL, RR, RL, Grip: findConstraints (Robot_Wrapper4.py:764-836)
  double* Cm = S.U + OFF_CM;
  double clb = 0.0, cub = 0.0;
  int prow = 0;
  if (cfg.con_com) {   // CoMConstraint (Robot_Wrapper4.py:669-694); EE_frame_pos[1] = FL, [2] = RR
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (lane < NV) Cm[(prow + r) * LDJ + lane] = jc[r];
      const double lo = ((S.pf[3 * 2 + r] - com[r]) / dt) * cfg.com_box_scale;
      const double hi = ((S.pf[3 * 1 + r] - com[r]) / dt) * cfg.com_box_scale;
      if (lane == prow + r) { clb = lo; cub = hi; }
    }
    prow += 2;
  }
  if (cfg.con_trunk) { // trunkConstraint (Robot_Wrapper4.py:707-754): LOCAL_WORLD_ALIGNED rows z, wx, wy, wz
    const bool sup = (lane < nv) && ((M.frame_support[WBC_FR_TRUNK] >> lane) & 1u);
    double wxp[3];
    cross3(ang, ptr, wxp);
    const double rowv[4] = {sup ? lin[2] + wxp[2] : 0.0, sup ? ang[0] : 0.0, sup ? ang[1] : 0.0, sup ? ang[2] : 0.0};
    const double* bc = A.in.trunk_box_center + (size_t)b * 4;
    // scipy as_euler('xyz') of the trunk rotation (:714-715)
    const double cur[4] = {ptr[2], atan2(Rtr[7], Rtr[8]), -asin(Rtr[6]), atan2(Rtr[3], Rtr[0])};
    const double var[4] = {bc[0] * cfg.trunk_box_z_frac, cfg.trunk_box_ang, cfg.trunk_box_ang, cfg.trunk_box_ang};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (lane < NV) Cm[(prow + r) * LDJ + lane] = rowv[r];
      const double lo = (((bc[r] - var[r]) - cur[r]) / dt) * cfg.trunk_box_scale;   // :735
      const double hi = (((bc[r] + var[r]) - cur[r]) / dt) * cfg.trunk_box_scale;   // :736
      if (lane == prow + r) { clb = lo; cub = hi; }
    }
    prow += 4;
  }
  for (int e = 0; e < WBC_NEE; ++e) {   // EEConstraint (Robot_Wrapper4.py:757-761): WORLD rows 0..2, 0 <= . <= 0
    if (!cfg.con_ee[e]) continue;
    const bool sup = (lane < nv) && ((M.frame_support[WBC_FR_EE0 + e] >> lane) & 1u);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      if (lane < NV) Cm[(prow + r) * LDJ + lane] = sup ? lin[r] : 0.0;
      if (lane == prow + r) { clb = 0.0; cub = 0.0; }
    }
    prow += 3;
  }
  // ---- velDamperJointConstraints (Robot_Wrapper4.py:572-637), index map from cfg
  double lb = 0.0, ub = 0.0;
  if (lane < nv) {
    if (!cfg.use_bounds) { lb = -1e30; ub = 1e30; }
    else {
      const double qi = S.qv[cfg.damper_qidx[ll]], lo = cfg.damper_lo[ll], hi = cfg.damper_hi[ll], vm = cfg.damper_vmax[ll];
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
  const QpResult res = qp_core(S, h, g, lb, ub, clb, cub, nv, A.prows, lane, ts);
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
    double R0[9], p0[3];
    quat_to_R(S.qv + 3, R0);
    p0[0] = S.qv[0]; p0[1] = S.qv[1]; p0[2] = S.qv[2];
    integrate_ff(S, lane, R0, p0, qn);
    if (lane >= 6 && lane < nv) qn[M.col_q[ll]] = S.qv[M.col_q[ll]] + v;
    if (lane >= nq && lane < NQ) qn[lane] = 0.0;
    WSYNC();
  }
#ifdef WBC_PROFILE
  STAMP(ts, T_END);
  if (A.prof && lane == 0 && res.status == WBC_QP_OPTIMAL) {
    for (int i = 1; i < T_N; ++i) atomicAdd(A.prof + i, ts[i] - ts[i - 1]);
    atomicAdd(A.prof + 0, 1ull);
    atomicAdd(A.prof + 8, (unsigned long long)res.iters);
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// kernels: single-wave workgroups, persistent over the batch (exit: b >= B, reached by every wave)
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(64, 2) wbc_tick_kernel(const KernelArgs A) {
  __shared__ Smem S;
  for (int b = blockIdx.x; b < A.B; b += gridDim.x) {
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));   // keep lane-derived values out of LICM's reach (they would be spilled)
    process_instance<MODE>(S, A, b, lane);
  }
}

// Stand-alone QP (QP_Wrapper.QP.solveQP): H, g (or A, b) and constraints straight from HBM.
__global__ void __launch_bounds__(64, 2) wbc_qp_kernel(const QpArgs A) {
  __shared__ Smem S;
  for (int b = blockIdx.x; b < A.B; b += gridDim.x) {
    int lane = threadIdx.x, n = A.n, p = A.p, m = A.m;
    asm volatile("" : "+v"(lane), "+s"(n), "+s"(p), "+s"(m));   // no LICM of lane/n-derived masks
    double h[NV];
    double g = 0.0;
    unsigned long long nullptr_ts[T_N];
    (void)nullptr_ts;
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
        }, h);
        g = (lane < n) ? -S.npv[li_clamp(lane)] : 0.0;
        WSYNC();
      } else {
        // vector path: lane k owns column k; row r of A is broadcast by uniform loads
#pragma unroll
        for (int i = 0; i < NV; ++i) h[i] = 0.0;
        for (int r = 0; r < m; ++r) {
          const double ak = (lane < n) ? Ab[(size_t)r * n + lane] : 0.0;
          g = fma(-ak, bb[r], g);
#pragma unroll
          for (int i = 0; i < NV; ++i) h[i] = fma((i < n) ? Ab[(size_t)r * n + i] : 0.0, ak, h[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) if (i == lane && lane >= n) h[i] = 1.0;
      if (A.H_out && lane < n) {
        double* o = A.H_out + (size_t)b * n * n + (size_t)lane * n;
#pragma unroll
        for (int k = 0; k < NV; ++k) if (k < n) o[k] = h[k];
      }
      if (A.g_out && lane < n) A.g_out[(size_t)b * n + lane] = g;
    } else {
      const double* Hb = A.H + (size_t)b * n * n;
#pragma unroll
      for (int k = 0; k < NV; ++k) h[k] = (lane < n && k < n) ? Hb[(size_t)lane * n + k] : ((k == lane) ? 1.0 : 0.0);
      g = (lane < n) ? A.g[(size_t)b * n + lane] : 0.0;
    }
    double* Cm = S.U + OFF_CM;
    for (int r = 0; r < p; ++r)
      if (lane < NV) Cm[r * LDJ + lane] = (lane < n) ? A.C[((size_t)b * p + r) * n + lane] : 0.0;
    const double lb = (lane < n) ? (A.lb ? A.lb[(size_t)b * n + lane] : -1e30) : 0.0;
    const double ub = (lane < n) ? (A.ub ? A.ub[(size_t)b * n + lane] : 1e30) : 0.0;
    const double clb = (lane < p) ? A.Clb[(size_t)b * p + lane] : 0.0;
    const double cub = (lane < p) ? A.Cub[(size_t)b * p + lane] : 0.0;
    WSYNC();
    const QpResult res = qp_core(S, h, g, lb, ub, clb, cub, n, p, lane, nullptr_ts);
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
  for (int b = blockIdx.x; b < A.B; b += gridDim.x) {
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));
    const DevModel& M = A.models[A.model_id ? A.model_id[b] : 0];
    const int nv = M.nv, nq = M.nq;
    if (lane < 32) S.qv[lane] = (lane < nq) ? A.q[(size_t)b * NQ + lane] : 0.0;
    const double v = (lane < nv) ? A.v[(size_t)b * NV + lane] * A.dt : 0.0;
    if (lane < 32) S.xv[lane] = v;
    WSYNC();
    double R0[9], p0[3];
    quat_to_R(S.qv + 3, R0);
    p0[0] = S.qv[0]; p0[1] = S.qv[1]; p0[2] = S.qv[2];
    double* qn = A.q_next + (size_t)b * NQ;
    integrate_ff(S, lane, R0, p0, qn);
    const int ll = lane & 31;
    if (lane >= 6 && lane < nv) qn[M.col_q[ll]] = S.qv[M.col_q[ll]] + v;
    if (lane >= nq && lane < NQ) qn[lane] = 0.0;
    WSYNC();
  }
}

static int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  (void)what;
  return e == hipSuccess ? 0 : (int)e;
}

int launch_tick(const KernelArgs& a, int mode, int grid, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (mode == MODE_TICK) hipLaunchKernelGGL(wbc_tick_kernel<MODE_TICK>, dim3(grid), dim3(64), 0, s, a);
  else if (mode == MODE_ASSEMBLE) hipLaunchKernelGGL(wbc_tick_kernel<MODE_ASSEMBLE>, dim3(grid), dim3(64), 0, s, a);
  else hipLaunchKernelGGL(wbc_tick_kernel<MODE_FK>, dim3(grid), dim3(64), 0, s, a);
  return check_launch("tick");
}
int launch_qp(const QpArgs& a, int grid, void* stream) {
  hipLaunchKernelGGL(wbc_qp_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, a);
  return check_launch("qp");
}
int launch_integrate(const IntegrateArgs& a, int grid, void* stream) {
  hipLaunchKernelGGL(wbc_integrate_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, a);
  return check_launch("integrate");
}
int tick_lds_bytes() { return (int)sizeof(Smem); }

}  // namespace wbc
