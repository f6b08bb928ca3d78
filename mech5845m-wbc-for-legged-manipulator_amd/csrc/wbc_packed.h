// wbc_packed.h — lane helpers and the LDS layout shared by the packed kernels (four instances per wavefront, lane = 16 r + s).
#pragma once
#include "wbc_common.h"

namespace wbc {

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

// The packed kernels' Cholesky sweep H' = L L' fused with the forward substitution L y = rhs (y comes in holding the lane's right-hand side: e_s, or g'
// on a padding lane), two columns per trip: the raw columns j and j + 1 of every row go through the LDS vectors c0v / c1v together and each lane redoes,
// for the rows below, the one update that column j + 1 receives from step j — the same operations in the same order as two single steps, one LDS round
// trip instead of two. Fully unrolled on fixed registers (round 4): trip j reads and updates the entries k >= j only — for N = 12: 42 b128 reads and
// 156 multiply-adds over the sweep instead of the 72 and 306 of the rotating-register loop this replaces (which did the rest on zeros: same results
// bit for bit); N = 16: 72 / 288 instead of 128 / 568. `wr`: this lane carries a row (s < N). Returns the smallest pivot (-1: not positive / NaN).
template <int N>
__device__ __forceinline__ double chol_sweep2(double (&h)[N], double (&y)[N], double* const c0v, double* const c1v, const int s, const bool wr) {
  double pmin = 1.0;
#pragma unroll
  for (int j = 0; j < N; j += 2) {
    WSYNC();
    if (wr) { c0v[s] = h[j]; c1v[s] = h[j + 1]; }
    WSYNC();
    double cm0[N], cm1[N];
#pragma unroll
    for (int k = j; k < N; k += 2) {
      const double2a v0 = lds2(c0v + k), v1 = lds2(c1v + k);
      cm0[k] = v0.x; cm0[k + 1] = v0.y; cm1[k] = v1.x; cm1[k + 1] = v1.y;
    }
    const double pj = cm0[j];
    pmin = (pj > 0.0) ? fmin(pmin, pj) : -1.0;
    const double rinv = rsqrt(pj), ipj = rinv * rinv;
    // step j on this row
    const double th = h[j] * ipj, ty = y[j] * ipj, yk = y[j] * rinv;
    const double h1 = fma(-th, cm0[j + 1], h[j + 1]), y1 = fma(-ty, cm0[j + 1], y[j + 1]);
    // step j as it acts on column j + 1 of the rows below (what their own lanes compute for themselves)
    const double a = cm0[j + 1];
#pragma unroll
    for (int k = j + 1; k < N; ++k) cm1[k] = fma(-(cm0[k] * ipj), a, cm1[k]);
    const double pj2 = cm1[j + 1];
    pmin = (pj2 > 0.0) ? fmin(pmin, pj2) : -1.0;
    const double rinv2 = rsqrt(pj2), ipj2 = rinv2 * rinv2;
    const double th2 = h1 * ipj2, ty2 = y1 * ipj2, yk2 = y1 * rinv2;
#pragma unroll
    for (int k = j + 2; k < N; ++k) h[k] = fma(-th2, cm1[k], fma(-th, cm0[k], h[k]));
#pragma unroll
    for (int k = j + 2; k < N; ++k) y[k] = fma(-ty2, cm1[k], fma(-ty, cm0[k], y[k]));
    y[j] = fma(-ty2, 0.0, yk); y[j + 1] = yk2;
  }
  return pmin;
}

// shared by the packed orth and box kernels: FK levels of their whole-tree schedule (DevPlan.q_fk) and the staged weights image wt [96]
constexpr int QLEV = 6;
static_assert(offsetof(WbcConfig, joint_w) - offsetof(WbcConfig, ee_W) == 84 * sizeof(double), "ee_W [30] ee_w [5] ee_gain [30] trunk [13] com_W [3] com_gain [3] joint_w");
constexpr int WT_W = 0, WT_w = 30, WT_G = 35, WT_CW = 78, WT_CG = 81;   // offsets inside wt

}  // namespace wbc
