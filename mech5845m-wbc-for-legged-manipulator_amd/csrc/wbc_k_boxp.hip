// wbc_k_boxp.hip — the packed box kernel wbc_tick_boxp_kernel<WARM>: task problems without constraint rows (the warm-up problem), four instances per wavefront.
#include "wbc_packed.h"

namespace wbc {

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
  const double pmin = chol_sweep2<16>(h, y, cl, yv, s, true);      // (wbc_packed.h)
  if (live && !(pmin > 0.0)) { status = WBC_QP_NUMERICAL; live = false; }
  XSTOP(6, y[0] + y[15] + h[15]);
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
    asm volatile("; WBC_TAIL_BEGIN" ::: "memory");   // (a comment in the assembly listing: tools/hot_path_spills.py cuts the control-flow graph here)
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

// One translation unit per PART (csrc/Makefile compiles this file once per part, in parallel): each part instantiates some of the kernel's
// variants; part 0 also holds the launcher and sees the other parts' variants as explicit-instantiation declarations.
#ifndef BOXP_PART
#define BOXP_PART -1      // -1: everything in one unit
#endif
#define KINST(...) template __global__ void wbc_tick_boxp_kernel<__VA_ARGS__>(const KernelArgs, const DevModel* __restrict__, const WbcConfig* __restrict__, const DevPlan* __restrict__);
#define KDECL(...) extern template __global__ void wbc_tick_boxp_kernel<__VA_ARGS__>(const KernelArgs, const DevModel* __restrict__, const WbcConfig* __restrict__, const DevPlan* __restrict__);
#if BOXP_PART == 0 || BOXP_PART == -1
KINST(false)
#endif
#if BOXP_PART == 1 || BOXP_PART == -1
KINST(true)
#elif BOXP_PART == 0
KDECL(true)
#endif
#undef KINST
#undef KDECL
#if BOXP_PART <= 0
int launch_tick_boxp(const KernelArgs& a, void* stream) {
  if (a.ws_in || a.ws_out) hipLaunchKernelGGL(wbc_tick_boxp_kernel<true>, dim3((a.B + 3) / 4), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  else hipLaunchKernelGGL(wbc_tick_boxp_kernel<false>, dim3((a.B + 3) / 4), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  return check_launch("tick_boxp");
}
#endif

}  // namespace wbc
