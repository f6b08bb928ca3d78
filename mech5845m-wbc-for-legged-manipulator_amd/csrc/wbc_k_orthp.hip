// wbc_k_orthp.hip — the packed orth kernel wbc_tick_orthp_kernel<INEQ, WARM>: task problems whose tasks touch the stance legs, four instances per wavefront.
#include "wbc_packed.h"

namespace wbc {

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
  // INEQ: the CoM box's rows and bounds (formed after the EE loop: their images must stay in AZ through the sweep) need the x / y rows of the CoM
  // Jacobian columns and the CoM again. Live across the EE loop those nine doubles were spilled around it — 76 B per lane out and back, 1.2 KB of
  // scratch traffic per tick each way (profiles/r03_pmc_summary_everything.txt: WRITE_SIZE 5x the outputs). They wait in LDS instead: the row
  // bounds' slots X[204..] are written only after the loop.
  double* const jst = I.X + 204;             // [16][4] jc0.x, jc0.y, jc1.x, jc1.y of lane s; then com.x, com.y
  if (INEQ) {
    sts2(jst + 4 * s, jc0[0], jc0[1]); sts2(jst + 4 * s + 2, jc1[0], jc1[1]);
    if (s == 0) sts2(jst + 64, com[0], com[1]);
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
      if (c_con_com) {
        const double2a j0 = lds2(jst + 4 * s), j1 = lds2(jst + 4 * s + 2);
        a0[4] = j0.x; a0[5] = j0.y; a1[4] = has1 ? j1.x : 0.0; a1[5] = has1 ? j1.y : 0.0;
      }
      block(a0, a1, zr, -1, true, true, false);          // (the images stay in AZ through the sweep, which leaves X alone)
    }
    const double2a comxy = lds2(jst + 64);                // (read before the row bounds take these slots)
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
        const double cr_ = r_ ? comxy.y : comxy.x;
        tl = ((I.pf[3 * 2 + r_] - cr_) * inv_dt) * cb_s;
        tu = ((I.pf[3 * 1 + r_] - cr_) * inv_dt) * cb_s;
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
  const double pmin = chol_sweep2<16>(h, y, I.cl, I.yv, s, true);      // (wbc_packed.h)
  QSTOP(6, y[0] + y[15] + h[15]);
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
        if (d < nv) { const int qi = M.col_q[d]; qn[qi] = qv[qi] + I.gp[d] * dt; }   // (the model's own q index of DoF d, as the other packed kernels read it)
      }
      if (s < NQ - nq) qn[nq + s] = 0.0;
    }
  }
  // ---- the tail: instances left out above (a flagged leg block; diagnostic orth_qr) are redone by this wave on the general path
  const unsigned long long tailm = __ballot(valid && defer && s == 0);
  if (tailm) {
    asm volatile("; WBC_TAIL_BEGIN" ::: "memory");   // (a comment in the assembly listing: tools/hot_path_spills.py cuts the control-flow graph here)
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

// One translation unit per PART (csrc/Makefile compiles this file once per part, in parallel): each part instantiates some of the kernel's
// variants; part 0 also holds the launcher and sees the other parts' variants as explicit-instantiation declarations.
#ifndef ORTHP_PART
#define ORTHP_PART -1      // -1: everything in one unit
#endif
#define KINST(...) template __global__ void wbc_tick_orthp_kernel<__VA_ARGS__>(const KernelArgs, const DevModel* __restrict__, const WbcConfig* __restrict__, const DevPlan* __restrict__);
#define KDECL(...) extern template __global__ void wbc_tick_orthp_kernel<__VA_ARGS__>(const KernelArgs, const DevModel* __restrict__, const WbcConfig* __restrict__, const DevPlan* __restrict__);
#if ORTHP_PART == 0 || ORTHP_PART == -1
KINST(false)
#endif
#if ORTHP_PART == 1 || ORTHP_PART == -1
KINST(true)
#elif ORTHP_PART == 0
KDECL(true)
#endif
#if ORTHP_PART == 2 || ORTHP_PART == -1
KINST(true, true)
#elif ORTHP_PART == 0
KDECL(true, true)
#endif
#undef KINST
#undef KDECL
#if ORTHP_PART <= 0
int launch_tick_orthp(const KernelArgs& a, void* stream, int ineq) {
  if (ineq && (a.ws_in || a.ws_out)) hipLaunchKernelGGL((wbc_tick_orthp_kernel<true, true>), dim3((a.B + 3) / 4), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  else if (ineq) hipLaunchKernelGGL(wbc_tick_orthp_kernel<true>, dim3((a.B + 3) / 4), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  else hipLaunchKernelGGL(wbc_tick_orthp_kernel<false>, dim3((a.B + 3) / 4), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  return check_launch("tick_orthp");
}
int orthp_lds_bytes() { return (int)(4 * sizeof(QInst)); }
#endif

}  // namespace wbc
