// wbc_k_sim3.hip — the compact one-instance sim3 kernel wbc_tick_sim3_kernel<WARM> and its deferred pass.
#include "wbc_common.h"

namespace wbc {

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

int sim3_lds_bytes() { return (int)sizeof(SmemC); }

}  // namespace wbc
