// wbc_k_sim3p.hip — the packed sim3 kernel wbc_tick_sim3p_kernel<WARM, TRUNK, QCON>: the benchmark path, four instances per wavefront.
#include "wbc_packed.h"

namespace wbc {

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
#ifndef SIM3P_WAVES
#define SIM3P_WAVES 2      // waves per SIMD the register allocation is made for (3: an experiment, tools/hot_path_spills.py with WBC_XFLAGS="-DSIM3P_WAVES=3 -DWBC_NO_TAIL")
#endif
template <bool WARM, bool TRUNK = false, bool QCON = false>
#ifdef SIM3P_NUM_VGPR
__attribute__((amdgpu_waves_per_eu(SIM3P_NUM_VGPR, SIM3P_NUM_VGPR)))
#endif
__global__ void __launch_bounds__(64, SIM3P_WAVES) wbc_tick_sim3p_kernel(const KernelArgs A, const DevModel* __restrict__ models,
                                                               const WbcConfig* __restrict__ cfgs, const DevPlan* __restrict__ plans) {
  // (the general kernel's layout shares the allocation: an instance this kernel cannot reduce — a stance-leg block of rank < 2 — is
  //  redone on the general path by the SAME wave at the end, see the tail; both layouts leave 8 waves per CU)
#ifdef SIM3P_NUM_VGPR      // (compile-only experiment: with the LDS size unknown to the compiler the occupancy attribute alone decides the register budget)
  extern __shared__ double dyn_lds_[];
  union SUU { SmemP P; Smem G; };
  SUU& SU = *reinterpret_cast<SUU*>(dyn_lds_);
#else
  __shared__ union { SmemP P; Smem G; } SU;
#endif
  static_assert(sizeof(Smem) <= 20480 && sizeof(SmemP) <= 20480, "8 waves per CU");
  SmemP& SP = SU.P;
  const int lane = threadIdx.x, r = lane >> 4, s = lane & 15, rbase = lane & 48;
  PInst& I = SP.I[r];
  PVec& V = SP.V[r];
  // (no loop over groups: a wave that takes two groups of four one after the other — half the workgroups, twice as long each — measured 6 % SLOWER
  //  even with a clean register allocation (opaque lane index + -mllvm -disable-machine-licm; without them the hoisted invariants spill 139 loads /
  //  91 stores into the packed path): profiles/r04_ab_licm_iters.txt. The first round of a launch costs twice a steady-state round — all waves
  //  stall on their inputs at once — and fewer, longer waves do not change that.)
  const int grp = (int)blockIdx.x;
  const int b_raw = 4 * grp + r;
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
  double at6[6] = {0, 0, 0, 0, 0, 0};      // TRUNK: the trunk task's image of base variable s (kept for the refinement's residual; its targets in V.pad_)
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
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) at6[rr] = at[rr];
    if (s < 6) {
      double vs = vel[0];
#pragma unroll
      for (int i = 1; i < 6; ++i) vs = (s == i) ? vel[i] : vs;
      V.pad_[s] = vs * trunk_w;              // b of the trunk rows (TrunkB, :914-920)
    }
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
  double bt[6] = {0, 0, 0, 0, 0, 0};        // the Grip rows' targets (b of qpb): kept for the refinement's residual, parked in V.in[28..33] below
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
      bt[i] = br;
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
      bt[3] = w3 * ee_w; bt[4] = w4 * ee_w; bt[5] = w5 * ee_w;
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
  double bp0 = 0.0, bp1 = 0.0;              // posture rows' targets (b of qpJointb) of reduced variable s / of leg DoF s: the refinement's residual
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
    bp0 = (s < n) ? (1.0 / nv) * u0 * joint_w : 0.0;
    bp1 = (s < nl) ? (1.0 / nv) * u1 * joint_w : 0.0;
  }
  if (s >= n) g = 0.0;
  if (A.post_static && P.post_pert) {     // the state qpJointb leaves behind: bounds and integrate see it
    WSYNC();
    if (((P.post_pert >> s) & 1u)) V.in[s] = (qv[s] + 0.0002) - (0.0002 * 2);
    if (16 + s < NQ && ((P.post_pert >> (16 + s)) & 1u)) V.in[16 + s] = (qv[16 + s] + 0.0002) - (0.0002 * 2);
  }
  WSYNC();
  if (s < 6) {                              // (every lane has read the Grip targets: their slots take the rows' b)
    double v = bt[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) v = (s == i) ? bt[i] : v;
    V.in[28 + s] = v;
  }
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
      // the Grip task's image in the new unknowns, A'' = A M (column j: a_j + u_j a_c*, column c*: wz a_c*; z2 moves no task frame): what the
      // refinement's residual is formed from at the end
      if (TRUNK) {                          // (the trunk task's image lives in registers: lanes s < 6)
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) {
          const double ac = bperm(at6[rr], rbase + cstar);
          if (on && s < n) at6[rr] = (s == cstar) ? wz * ac : fma(us, ac, at6[rr]);
        }
      }
      if (QCON) {                           // (the second kinematics pass took the image's place in M2: this variant keeps it in registers)
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) {
          const double ac = bperm(a[rr], rbase + cstar);
          if (on && s < n) a[rr] = (s == cstar) ? wz * ac : fma(us, ac, a[rr]);
        }
      } else {
        const double2a c0 = lds2(At + cstar * 6), c1 = lds2(At + cstar * 6 + 2), c2 = lds2(At + cstar * 6 + 4);
        WSYNC();
        if (on && s < n) {
          const double2a m0 = lds2(At + s * 6), m1 = lds2(At + s * 6 + 2), m2 = lds2(At + s * 6 + 4);
          const bool pv = s == cstar;
          sts2(At + s * 6, pv ? wz * c0.x : fma(us, c0.x, m0.x), pv ? wz * c0.y : fma(us, c0.y, m0.y));
          sts2(At + s * 6 + 2, pv ? wz * c1.x : fma(us, c1.x, m1.x), pv ? wz * c1.y : fma(us, c1.y, m1.y));
          sts2(At + s * 6 + 4, pv ? wz * c2.x : fma(us, c2.x, m2.x), pv ? wz * c2.y : fma(us, c2.y, m2.y));
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
        const double pB = bperm(bp1, rbase + (l2 & 15)), pA = bperm(bp0, rbase + cstar);   // the posture rows' targets follow their DoF
        if (on && s == cstar) { lb = nl_; ub = nu_; dofA = dB; bp0 = pB; }
        if (on && s == p_keep + l2) { clb = ol_; cub = ou_; }
        if (on && s == l2) { dofB = dA; bp1 = pA; }
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
  // ---- Cholesky H' = L L' fused with the substitution L y = e_s (two columns per trip on fixed registers; columns broadcast through V.cl / V.yv)
  WSYNC();
  V.cl[s] = 0.0; V.cl[16 + s] = 0.0;        // (the row bounds were staged there)
  V.yv[s] = 0.0; V.tv[s] = 0.0;             // second column vector of the blocked sweep: yv | tv, 32 contiguous entries, zero tail
  double y[PV];
#pragma unroll
  for (int k = 0; k < PV; ++k) y[k] = (k == s) ? 1.0 : 0.0;
  const double pmin = chol_sweep2<PV>(h, y, V.cl, V.yv, s, s < PV);      // (wbc_packed.h)
  if (live && !(pmin > 0.0)) { status = WBC_QP_NUMERICAL; live = false; }
  PSTOP(4, y[0] + y[11] + h[11]);
  // y = row s of J0 = L^-T.  jf2 = |J0|_F^2 per instance
  double sq = 0.0;
#pragma unroll
  for (int k = 0; k < PV; ++k) sq = fma(y[k], y[k], sq);
  const double jf2 = rsum16(s < PV ? sq : 0.0);
  double* const J = I.M2;
  double* const T = I.M1;
  if (!QCON) {                              // the Grip image of reduced variable s comes back from LDS before J takes its place (free registers across the sweep)
    const double2a t0 = lds2(At + s * 6), t1 = lds2(At + s * 6 + 2), t2 = lds2(At + s * 6 + 4);
    a[0] = t0.x; a[1] = t0.y; a[2] = t1.x; a[3] = t1.y; a[4] = t2.x; a[5] = t2.y;
  }
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
  // ================================ iterative refinement ==========================================
  // One step at the final working set (QP_Wrapper.py:37 asks qpOASES for numRefinementSteps = 100; oracle: qp_refine). With the slots' normals
  // n_k, right-hand sides b_k and multipliers u_k:   r1 = -(grad f(y) - sum u_k n_k),  r2_k = b_k - n_k'y,   dy = J1 T' r2 + J2 J2' r1.
  // grad f comes from the UNFACTORED least-squares data — the Grip image A Z (a[] of every lane), the posture rows d (y_s - u_s) of the reduced
  // variables and of the eliminated leg DoF (x_l = G_l y: rows p_keep + l of Cq) — never from H': fl(H') carries the 1.5e-9 posture block
  // with 1e-5 relative error, which IS the 1e-6 the plain method is off by (cond(H) ~ 3e9); the least-squares residual's own rounding lies in
  // the range of (A Z)', where H' is well conditioned.
#ifdef WBC_ABLATE
  const bool skip_refine_ = A.dbg_stop == 109;                        // timing cut: the whole tick without the refinement step
#else
  const bool skip_refine_ = false;
#endif
  if (A.refine > 0 && !skip_refine_) {
    // r2 — the active constraints' own residual b_k - n_k'y — is formed on the WARM variant only (its iterate is rebuilt from the factors after the
    // seeds: refresh). On the cold path the dual method's steps keep the working set satisfied to rounding and the term changes nothing (same-box
    // check: 6.68e-9 worst with and without it).
    WSYNC();
    V.xv[s] = has_b ? x : 0.0;
    V.yv[s] = 0.0;
    if (WARM) { V.cl[s] = lb; V.cl[16 + s] = ub; V.tv[s] = clb; V.dv[s] = cub; }
    WSYNC();
    const double2a x0 = lds2(V.xv), x1_ = lds2(V.xv + 2), x2 = lds2(V.xv + 4);
    auto rowval = [&](const int rr_) -> double {     // C'_rr y (every reduced row has base support only)
      const double2a c0 = lds2(I.Cq + rr_ * 6), c1 = lds2(I.Cq + rr_ * 6 + 2), c2 = lds2(I.Cq + rr_ * 6 + 4);
      return fma(c0.x, x0.x, fma(c0.y, x0.y, fma(c1.x, x1_.x, fma(c1.y, x1_.y, fma(c2.x, x2.x, c2.y * x2.y)))));
    };
    const double vrow = has_r ? rowval(s) : 0.0;     // row s at y; rows p_keep + l: the velocity of leg DoF l
    // slot s: the signed multiplier of its constraint
    const int cc = a_code & 255, sd = (a_code >> 8) & 1;
    const bool slot = s < q, srow = slot && cc >= n;
    const int rr_ = srow ? ((cc - n) & 15) : 0, iv = cc & 15;
    const double sgn = sd ? -1.0 : 1.0;
    const double us = slot ? sgn * u : 0.0;
    double r2 = 0.0;
    if (WARM) {
      const double vbd = V.xv[iv];
      const double val = srow ? rowval(rr_) : vbd;
      const double* const bsrc = srow ? ((sd ? V.dv : V.tv) + rr_) : (V.cl + (sd ? 16 : 0) + iv);
      r2 = slot ? sgn * (*bsrc - val) : 0.0;
    }
    const double bpl = bperm(bp1, rbase + ((s - p_keep) & 15));     // posture target of the leg DoF whose velocity row s carries
    const double wl = (s >= p_keep && s < p) ? dpost * (bpl - dpost * vrow) : 0.0;   // the leg DoF's posture residual: row s's weight in r1
    WSYNC();
    V.xv[s] = wl;
    if (srow) V.yv[rr_] = us;                                           // the active rows' multipliers (yv was cleared above)
    if (slot && !srow) V.dv[iv] = us;                                   // active bounds' multipliers by variable (read where actm says so)
    if (WARM) V.cl[s] = r2;
    // Grip rows: e = b - (A Z) y, then (A Z)'e
    double r1 = 0.0;
    {
      const double2a b0 = lds2(V.in + 28), b1 = lds2(V.in + 30), b2 = lds2(V.in + 32);
      const double btv[6] = {b0.x, b0.y, b1.x, b1.y, b2.x, b2.y};
      const double xm = has_b ? x : 0.0;
      // (the products are made opaque: fused into the butterfly's first add — a x + neighbour's rounded product — the two lanes of a pair end with
      //  sums one rounding apart, e differs from lane to lane, and r1 = sum a e leaves the range of (A Z)': 5e-7 on q̇ instead of 7e-9, measured)
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) {
        double pr = a[rr] * xm;
        asm volatile("" : "+v"(pr));
        r1 = fma(a[rr], btv[rr] - rsum16(pr), r1);
      }
      if (TRUNK) {                            // trunk rows (base columns): the same with their image and targets
        const double2a t0 = lds2(V.pad_), t1 = lds2(V.pad_ + 2), t2 = lds2(V.pad_ + 4);
        const double ttv[6] = {t0.x, t0.y, t1.x, t1.y, t2.x, t2.y};
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) {
          double pr = at6[rr] * xm;
          asm volatile("" : "+v"(pr));
          r1 = fma(at6[rr], ttv[rr] - rsum16(pr), r1);
        }
      }
    }
    r1 = fma(dpost, bp0 - dpost * x, r1);
    WSYNC();
    if (actm & 1) r1 += V.dv[s];
    if (s < 6) {
#pragma unroll
      for (int rr = 0; rr < PN; rr += 2) {
        const double2a m2 = lds2(V.yv + rr), l2 = lds2(V.xv + rr);
        r1 = fma((rr < p) ? I.Cq[rr * 6 + s] : 0.0, m2.x + l2.x, r1);
        r1 = fma((rr + 1 < p) ? I.Cq[(rr + 1) * 6 + s] : 0.0, m2.y + l2.y, r1);
      }
    }
    if (!has_b) r1 = 0.0;
    V.tv[s] = r1;
    WSYNC();
    const int sv = s < PV ? s : PV - 1;
    double dy1 = 0.0, dy2 = 0.0;
#pragma unroll
    for (int j = 0; j < PV; j += 2) {
      const double2a r1v = lds2(V.tv + j);
      dy2 = fma(J[j * PLD + sv], r1v.x, fma(J[(j + 1) * PLD + sv], r1v.y, dy2));       // J'r1
      if (WARM) {
        const double2a r2v = lds2(V.cl + j);
        dy1 = fma(T[j * PLD + sv], r2v.x, fma(T[(j + 1) * PLD + sv], r2v.y, dy1));     // T'r2 (T is zero outside the slots)
      }
    }
    V.yv[s] = (s < PV) ? ((s < q) ? dy1 : dy2) : 0.0;                                  // (every lane read yv before the last fence)
    WSYNC();
    double da = 0.0, db = 0.0;
#pragma unroll
    for (int k = 0; k < PV; k += 2) { const double2a j2 = lds2(J + sv * PLD + k), w2 = lds2(V.yv + k); da = fma(j2.x, w2.x, da); db = fma(j2.y, w2.y, db); }
    // the correction is small against x (1e-6 on the tick, up to 1e-3 on a cond-1e10 problem); one that is not (> 0.25 max(1, |x|)) or is non-finite — a working set on the edge of dependence — is not applied (oracle: same rule)
    const double dxl = has_b ? da + db : 0.0;
    const double dmax = -rmin16(-fabs(dxl)), xmax = fmax(1.0, -rmin16(has_b ? -fabs(x) : 0.0));
    const bool nanr = ((__ballot(dxl != dxl) >> rbase) & 0xFFFFull) != 0;
    if (has_b && status == WBC_QP_OPTIMAL && !nanr && dmax <= 0.25 * xmax) x += dxl;
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
    asm volatile("; WBC_TAIL_BEGIN" ::: "memory");   // (a comment in the assembly listing: tools/hot_path_spills.py cuts the control-flow graph here)
#pragma unroll 1
    for (int rr = 0; rr < 4; ++rr) {
      if (!((tailm >> (16 * rr)) & 1ull)) continue;
      tail_instance<WARM, false>(&SU.G, 4 * grp + rr, models, cfgs, plans);
    }
  }
#endif
}

// One translation unit per PART (csrc/Makefile compiles this file once per part, in parallel): each part instantiates some of the kernel's
// variants; part 0 also holds the launcher and sees the other parts' variants as explicit-instantiation declarations.
#ifndef SIM3P_PART
#define SIM3P_PART -1      // -1: everything in one unit
#endif
#define KINST(...) template __global__ void wbc_tick_sim3p_kernel<__VA_ARGS__>(const KernelArgs, const DevModel* __restrict__, const WbcConfig* __restrict__, const DevPlan* __restrict__);
#define KDECL(...) extern template __global__ void wbc_tick_sim3p_kernel<__VA_ARGS__>(const KernelArgs, const DevModel* __restrict__, const WbcConfig* __restrict__, const DevPlan* __restrict__);
#if SIM3P_PART == 0 || SIM3P_PART == -1
KINST(false, false)
#endif
#if SIM3P_PART == 1 || SIM3P_PART == -1
KINST(true, false)
KINST(false, true)
KINST(true, true)
#elif SIM3P_PART == 0
KDECL(true, false)
KDECL(false, true)
KDECL(true, true)
#endif
#if SIM3P_PART == 2 || SIM3P_PART == -1
KINST(false, false, true)
KINST(true, false, true)
#elif SIM3P_PART == 0
KDECL(false, false, true)
KDECL(true, false, true)
#endif
#undef KINST
#undef KDECL
#if SIM3P_PART <= 0
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
int sim3p_lds_bytes() { return (int)sizeof(SmemP); }
#endif

}  // namespace wbc
