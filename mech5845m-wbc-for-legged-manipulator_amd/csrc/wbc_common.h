// wbc_common.h — shared device code of the gfx950 (MI355X, CDNA4) kernels of the batched whole-body-control tick (included by every wbc_k_*.hip).
#pragma once
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
#include <type_traits>
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
  double pft[16];                       // the five EE frame origins at the TASK state (pf follows a second kinematics pass): the refinement's residual
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
// RF: the caller's residual functor for the iterative refinement at the final working set (qp_refine below): rf(x) returns, on lane k < n,
// entry k of -grad f(x) formed from the caller's UNFACTORED data (least-squares data where it has them); every lane calls it (it may reduce
// over the wave) and it may use S.xv / S.yv / S.dv / S.npv. NoRefine: nothing is compiled in.
// On the stand-alone QP boundary (arbitrary problems) the refinement runs where it has something to repair: the plain dual method's error is
// ~cond(H) eps |x|, so a problem whose smallest Cholesky pivot is above WBC_REFINE_COND x its largest diagonal entry (cond(H) of ~1e5: error < 1e-9)
// keeps the plain answer (`adaptive`). The tick kernels refine always: on their REDUCED problems the pivot ratio is no usable estimate — a stance
// near a singular leg block inflates the base block through G'G while the posture-only directions stay at 1e-9 (soak: 2.3e-7 left unrepaired
// with the test on, 4.7e-8 without). The oracle refines always.
constexpr double WBC_REFINE_COND = 1e-5;
struct NoRefine {
  static constexpr bool enabled = false;
  __device__ NoRefine() {}
  template <class F> __device__ NoRefine(const F&) {}
  __device__ double operator()(double) const { return 0.0; }
};
template <class F> struct Refine { static constexpr bool enabled = true; const F& f; __device__ __forceinline__ double operator()(double v) const { return f(v); } };

// One step of iterative refinement at the final working set (QP_Wrapper.py:37 asks qpOASES for numRefinementSteps = 100; oracle: qp_refine).
// Needs nothing of the dual method but its final J (J J' = H^-1, the first q columns spanning the active normals) and the list of
// active constraints: R = J1'N is REBUILT from them (one product per slot), so equality slots — whose part of R the register-resident QR
// never stores — are corrected like the others, and T = R^-1 (inequality slots only) is not used; R takes T's place in RA.
//     gneg = -grad f(x) (rf),  u = -R^-1 J1'gneg,  r1 = gneg + N'u,  r2_k = b_k - n_k'x,   x += J1 R^-T r2 + J2 J2' r1.
// Forming N'u before the product with J2 matters: J2'gneg alone is a sum of O(|J| |grad f|) terms that cancel to ~0 (|J| ~ 1/d = 2.6e4 on the
// benchmark tick) and its rounding would come back through J2 as 1e-7 on x; r1 is small, so its products are clean.
template <int NM, class SM, int CS, class RF>
__device__ __forceinline__ double qp_refine(SM& S, const RF& rf, const double x_in, const int n, const int p, const int lane, const int q,
                                            const int a_code, const bool fixb, const double lb, const double ub, const double clb, const double cub) {
  const int li = lane < NM ? lane : NM - 1;
  double* const R = S.RA;
  const double* const J = S.RB;
  const double* const Cm = S.RC;
  const double gneg_ = rf(x_in);
  const double gneg = (lane < n && !fixb) ? gneg_ : 0.0;
  WSYNC();
  if (lane < 32) { S.xv[lane] = (lane < n) ? x_in : 0.0; S.dv[lane] = lb; S.yv[lane] = ub; S.npv[lane] = clb; S.lv[lane] = cub; S.dinv[lane] = gneg; }
  WSYNC();
  // slot `lane`: its constraint's residual b - n'x
  const int cc = a_code & 255, sd = (a_code >> 8) & 1;
  const bool slot = lane < q, srow = cc >= n;
  const int rr = (srow && slot) ? cc - n : 0, iv = (slot && !srow) ? cc : 0;
  double val = S.xv[iv & 31];
  if (__ballot(slot && srow)) {
    double v = 0.0, vb = 0.0;
#pragma unroll
    for (int k = 0; k < NM; k += 2) { const double2a c2 = lds2(Cm + rr * CS + k); const double2a x2 = lds2(S.xv + k); v = fma(c2.x, x2.x, v); vb = fma(c2.y, x2.y, vb); }
    if (srow) val = v + vb;
  }
  const double bnd = srow ? (sd ? S.lv[rr & 31] : S.npv[rr & 31]) : (sd ? S.yv[iv & 31] : S.dv[iv & 31]);
  const double sgn_s = sd ? -1.0 : 1.0;
  double t = slot ? sgn_s * (bnd - val) : 0.0;          // r2 of slot `lane`
  // w = J'gneg (lane k: column k of J)
  double w = 0.0;
#pragma unroll
  for (int i = 0; i < NM; ++i) w = fma(J[i * LDJ + li], S.dinv[i], w);
  WSYNC();
  // R = J1'N, column by column (slot k uniform): lane i <= k keeps R[i][k]; what rounding leaves below the diagonal is dropped
#pragma unroll 1
  for (int k = 0; k < q; ++k) {
    const int ck = rdli(a_code, k);
    const int ip = ck & 255;
    const double sg = (ck >> 8) ? -1.0 : 1.0;
    double d;
    if (ip >= n) {
      const int r_ = ip - n;
      double d0 = 0.0, d1 = 0.0;
#pragma unroll
      for (int i = 0; i < NM; i += 2) {
        const double2a c2 = lds2(Cm + r_ * CS + i);
        d0 = fma(J[i * LDJ + li], c2.x, d0); d1 = fma(J[(i + 1) * LDJ + li], c2.y, d1);
      }
      d = (d0 + d1) * sg;
    } else d = sg * J[ip * LDJ + li];
    if (lane < q) R[lane * LDJ + k] = (lane <= k) ? d : 0.0;
  }
  WSYNC();
  // multipliers: R u = -w1 (back substitution; lane k ends up with u_k)
  double c = slot ? -w : 0.0, um = 0.0;
#pragma unroll 1
  for (int k = q - 1; k >= 0; --k) {
    const double uk = rdl(c, k) / R[k * LDJ + k];
    if (lane == k) um = uk;
    if (lane < k) c = fma(-R[lane * LDJ + k], uk, c);
  }
  // r1 = gneg + N'u
  double r1 = gneg;
#pragma unroll 1
  for (int k = 0; k < q; ++k) {
    const int ck = rdli(a_code, k);
    const int ip = ck & 255;
    const double uk = rdl(um, k) * ((ck >> 8) ? -1.0 : 1.0);
    if (ip >= n) r1 = fma(uk, Cm[(ip - n) * CS + li], r1);
    else if (lane == ip) r1 += uk;
  }
  if (!(lane < n) || fixb) r1 = 0.0;
  if (lane < 32) S.dinv[lane] = r1;
  WSYNC();
  double dy = 0.0;
#pragma unroll
  for (int i = 0; i < NM; ++i) dy = fma(J[i * LDJ + li], S.dinv[i], dy);     // J'r1: lanes >= q keep it (J2'r1)
  // R'dy1 = r2 (forward substitution; lane k ends up with dy1_k)
  double dy1 = 0.0;
#pragma unroll 1
  for (int k = 0; k < q; ++k) {
    const double dk = rdl(t, k) / R[k * LDJ + k];
    if (lane == k) dy1 = dk;
    if (lane > k && lane < q) t = fma(-R[k * LDJ + lane], dk, t);
  }
  WSYNC();
  if (lane < 32) S.dinv[lane] = (lane < q) ? dy1 : ((lane < NM) ? dy : 0.0);
  WSYNC();
  double da = 0.0, db = 0.0;
#pragma unroll
  for (int k = 0; k < NM; k += 2) { const double2a j2 = lds2(J + li * LDJ + k); const double2a w2 = lds2(S.dinv + k); da = fma(j2.x, w2.x, da); db = fma(j2.y, w2.y, db); }
  WSYNC();
  // the correction is small against x (1e-6 on the tick, up to 1e-3 on a cond-1e10 problem); one that is not (> 0.25 max(1, |x|)) or is non-finite — a working set on the edge of dependence — is not applied (oracle: same rule)
  const double dxl = (lane < n && !fixb) ? da + db : 0.0;
  const double dmax = -wmin(lane < 32 ? -fabs(dxl) : 0.0), xmax = fmax(1.0, -wmin(lane < 32 ? -fabs(lane < n ? x_in : 0.0) : 0.0));
  const bool sane = __ballot(dxl != dxl) == 0 && dmax <= 0.25 * xmax;
  return sane ? x_in + dxl : x_in;
}

template <int NM, class SM = Smem, int CS = LDJ, bool WARM = false, class RF = NoRefine>
__device__ __forceinline__ QpResult qp_core(SM& S, const double g_in, const double lb_in, const double ub_in,
                                            const double clb_in, const double cub_in, const int n, const int p, const int lane,
                                            unsigned long long* ts, const int dbg_stop = 0, const int ws_b_in = 0, const int ws_r_in = 0,
                                            const RF& rf = RF(), const int refine = 0, const bool adaptive = false) {
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
  // largest diagonal entry of H: with the smallest Cholesky pivot it tells whether the refinement has anything to repair (WBC_REFINE_COND)
  double hmax = 0.0;
  if (RF::enabled) {
    double hd = 0.0;
#pragma unroll
    for (int k = 0; k < NM; ++k) hd = (k == li) ? h[k] : hd;
    hmax = -wmin((lane < NM && lane < 32) ? -hd : 0.0);
  }

  // ---- Cholesky H = L L', right-looking, the row in registers; column j is broadcast through S.cl; L itself is never stored (the
  // substitutions ride along).
  double pmin = 1.0;
  // Forward substitutions L y = rhs, one right-hand side per lane:
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
#pragma unroll
  for (int j = 0; j < NM; ++j) {
    const double pj = rdl(h[j], j);
    pmin = (pj > 0.0) ? fmin(pmin, pj) : -1.0;            // (a NaN pivot must fail the test below; fmin would drop it)
    const double rinv = rsqrt(pj);
    const double l = h[j] * rinv;
    WSYNC();
    if (lane < NM) S.cl[lane] = l;
    WSYNC();
    double cm[NM];
#pragma unroll
    for (int k = (j + 1) & ~1; k < NM; k += 2) { const double2a v = lds2(S.cl + k); cm[k] = v.x; cm[k + 1] = v.y; }
    const double yk = y[j] * rinv;
#pragma unroll
    for (int k = j + 1; k < NM; ++k) { h[k] = fma(-l, cm[k], h[k]); y[k] = fma(-cm[k], yk, y[k]); }
    y[j] = yk;
    // every value of the step is pinned to a register here: left free, the scheduler spreads the unrolled steps over each other and the
    // allocator pays with 600-1300 spilled VGPRs (the packed kernels' sweeps, the same code shape, allocate cleanly without)
#pragma unroll
    for (int k = j + 1; k < NM; ++k) asm volatile("" : "+v"(h[k]), "+v"(y[k]));
  }
  STAMP(ts, T_CHOL);
  if (!(pmin > 0.0)) { res.status = WBC_QP_NUMERICAL; return res; }
  if (dbg_stop == 6) { res.x = y[0] + h[NM - 1]; return res; }      // ablation timing: fused Cholesky / substitution sweep done

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
        int ecode = 0;                                   // (equality slots: kept for the refinement, which rebuilds R from the active constraints)
        if (mb) { const int c = ctz64(mb); mb &= mb - 1; col = S.RB + c * LDJ; be = rdl(lb, c); n2 = 1.0; ecode = c; }
        else if (mr) { const int c = ctz64(mr); mr &= mr - 1; col = S.RA + c * LDJ; be = rdl(clb, c); n2 = rdl(cn2, c); ecode = n + c; }
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
        if (lane == 0) { S.dinv[e] = be; S.lv[e] = n2; if (RF::enabled) S.xv[e] = (double)ecode; }   // (dinv / lv / xv are free after the substitution)
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
      if (RF::enabled && !seed && lane == q) a_code = (int)S.xv[e];
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
  if (RF::enabled) {
    // (`adaptive`: a well-conditioned problem — smallest pivot above WBC_REFINE_COND x the largest diagonal entry — skips the step)
    if (refine > 0 && res.status == WBC_QP_OPTIMAL && (!adaptive || pmin < WBC_REFINE_COND * hmax))
      x = qp_refine<NM, SM, CS, RF>(S, rf, x, n, p, lane, q, a_code, fixb, lb, ub, clb, cub);
  }
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
// RFULL: the caller's full-space residual functor (lane d: entry d of -grad f at the velocity it is handed on lane d) for the refinement; here it is
// wrapped into the reduced coordinates: x = Z y, r' = Z'r.
template <bool WARM = false, class KA = KernelArgs, class RFULL = NoRefine>
__device__ __forceinline__ bool contact_presolve(Smem& S, const KA& A, const DevModel& M, const WbcConfig& cfg,
                                                 const DevPlan& P, const double dpost, const double g, const double lb,
                                                 const double ub, const double clb, const double cub, const int lane,
                                                 unsigned long long* ts, QpResult& res, const unsigned long long ws0 = 0ull,
                                                 const unsigned long long ws1 = 0ull, const RFULL& rfull = RFULL(), const int refine = 0) {
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
  // the refinement's residual in the reduced coordinates: y -> x = Z y by DoF -> the caller's full-space residual -> Z'r
  auto resid_red = [&](const double yk) -> double {
    WSYNC();
    if (lane < 32) S.xv[lane] = (lane < n_red) ? yk : 0.0;
    WSYNC();
    double xd = 0.0;
    if (my_pos >= 0) xd = S.xv[my_pos];
    else if (my_l >= 0) {
#pragma unroll
      for (int c = 0; c < 6; ++c) xd = fma(Gm[my_l * GS + c], S.xv[c], xd);
    }
    const double rd = rfull((lane < nv) ? xd : 0.0);
    WSYNC();
    if (lane < 32) S.yv[lane] = (lane < nv) ? rd : 0.0;
    WSYNC();
    double rk = (lane < n_red) ? S.yv[fj] : 0.0;
    if (lane < 6) {
#pragma unroll
      for (int l = 0; l < 12; ++l) rk = fma((l < nl) ? Gm[l * GS + lane] : 0.0, S.yv[legd[l]], rk);
    }
    WSYNC();
    return rk;
  };
  typedef typename std::conditional<RFULL::enabled, Refine<decltype(resid_red)>, NoRefine>::type RRED;
  const RRED rred{resid_red};
  if (n_red <= 12) res = qp_core<12, Smem, LDJ, WARM, RRED>(S, g_red, lb_red, ub_red, nclb, ncub, n_red, i2, lane, ts, 0, sd_b, sd_r, rred, refine);   // (qp_core's sweeps cost ~NM^2)
  else res = qp_core<NR, Smem, LDJ, WARM, RRED>(S, g_red, lb_red, ub_red, nclb, ncub, n_red, i2, lane, ts, 0, sd_b, sd_r, rred, refine);
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
  // the lane's WORLD Jacobian column at the TASK state, parked for the refinement's residual (a second kinematics pass below replaces lin / ang):
  // the Cartesian targets [28..57] and the input groups 2 and 3 [64..191] of the staging image are dead from here on
  double* const colst = S.in + ((lane < 21) ? 64 + 6 * lane : 28 + 6 * ((lane < NV ? lane : NV - 1) - 21));
  if (MODE == MODE_TICK && lane < NV) { sts2(colst, lin[0], lin[1]); sts2(colst + 2, lin[2], ang[0]); sts2(colst + 4, ang[1], ang[2]); }
  if (MODE == MODE_TICK && lane < 15) S.pft[lane] = S.pf[lane];
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
  // The refinement's residual in full space (lane d = DoF d): -grad f(x) = A'(b - A x) from the UNFACTORED task stack, block by block as pass 1
  // formed it — the lane's column of every Cartesian block from its parked Jacobian column, the frame origins S.pf and the configuration's weights,
  // e = b - A x by wave reductions against the targets S.bt, then the posture rows. Nothing of H enters (qp_refine says why). The CoM task's
  // rows need the CoM Jacobian column too, which is not kept: those configurations (cond(H) ~ 1e6: 3e-8 without) are not refined here.
  auto resid_full = [&](const double xd) -> double {
    const double2a s0 = lds2(colst), s1 = lds2(colst + 2), s2 = lds2(colst + 4);
    const double lt[3] = {s0.x, s0.y, s1.x}, at[3] = {s1.y, s2.x, s2.y};
    double rr_ = 0.0;
    int row_ = 0;
#pragma unroll 1
    for (unsigned tm = P.task_ee_mask; tm; tm &= tm - 1) {
      const int e = __ffs((int)tm) - 1;
      const bool sup = (lane < nv) && ((M.frame_support[WBC_FR_EE0 + e] >> lane) & 1u);
      const double w = cfg.ee_w[e];
      const double pfe[3] = {S.pft[3 * e], S.pft[3 * e + 1], S.pft[3 * e + 2]};
      double wxp[3];
      cross3(at, pfe, wxp);
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const double a = sup ? cfg.ee_W[e][r] * ((r < 3 ? lt[r < 3 ? r : 0] + wxp[r < 3 ? r : 0] : at[r < 3 ? 0 : r - 3]) * w) : 0.0;
        const double er = S.bt[row_ + r] - wsum(a * xd);
        rr_ = fma(a, er, rr_);
      }
      row_ += 6;
    }
    if (cfg.task_trunk) {
      const bool sup = (lane < nv) && ((M.frame_support[WBC_FR_TRUNK] >> lane) & 1u);
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const double a = sup ? (cfg.trunk_W[r] * (r < 3 ? lt[r < 3 ? r : 0] : at[r < 3 ? 0 : r - 3])) * cfg.trunk_w : 0.0;
        const double er = S.bt[row_ + r] - wsum(a * xd);
        rr_ = fma(a, er, rr_);
      }
      row_ += 6;
    }
    if (lane < nv) rr_ = fma(dpost, upost - dpost * xd, rr_);
    return (lane < nv) ? rr_ : 0.0;
  };
  typedef Refine<decltype(resid_full)> RFULL;
  const RFULL rfull{resid_full};
  const int n_refine = cfg.task_com ? 0 : A.refine;
  if (!contact_presolve<WARM, KA, RFULL>(S, A, M, cfg, P, dpost, g, lb, ub, clb, cub, lane, ts, res, w0, w1, rfull, n_refine) &&
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
      res = qp_core<24, Smem, LDJ, false, RFULL>(S, (lane < n_eff) ? g : 0.0, lb, ub, clb, cub, n_eff, A.prows, lane, ts, 0, 0, 0, rfull, n_refine);
      res.iters += ntail;
      if (lane >= n_eff) res.x = 0.0;
    } else
    res = qp_core<NV, Smem, LDJ, WARM, RFULL>(S, g, lb, ub, clb, cub, nv, A.prows, lane, ts, 0, sb == 3 ? 0 : sb, sr == 3 ? 0 : sr, rfull, n_refine);
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

// (the kernels live in wbc_k_*.hip, one translation unit per kernel family: they compile in parallel)

// One instance on the general path, called by the packed kernels for what they cannot reduce themselves (the TAIL: a stance-leg block of rank < 2,
// a flagged block on the orth kernel). It reads the kernel's argument block AGAIN, through the kernarg segment pointer (KernelArgs is the first
// kernel parameter of both callers): handed the caller's own `A`, the general path's ~90 scalars were fetched at kernel entry and kept alive —
// spilled to VGPR lanes — across the whole packed path: 480 extra v_writelane / v_readlane in the common path, 3 % of the step (same-box A/B,
// tools/ab_bench.sh). A real call is not an option: arguments arrive in VGPRs, and the general path pins configuration scalars to SGPRs.
template <bool WARM, bool ORTH>
__device__ __forceinline__ void tail_instance(Smem* Sp, const int bt_v, const DevModel* __restrict__ models, const WbcConfig* __restrict__ cfgs,
                                              const DevPlan* __restrict__ plans) {   // (the kernel's own noalias table pointers: scalar loads)
  Smem& S = *Sp;
  asm volatile("; WBC_TAIL_BEGIN" ::: "memory");                 // (a comment in the assembly listing: tools/hot_path_spills.py splits the ISA here)
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
  asm volatile("; WBC_TAIL_END" ::: "memory");
}

static int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  (void)what;
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace wbc
