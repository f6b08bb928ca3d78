// wbc_device.h — host-visible description of what the gfx950 kernels consume (internal, C++).
#pragma once
#include <stdint.h>
#include "../../include/wbc.h"

namespace wbc {

constexpr int NV = WBC_V_STRIDE;   // 26: QP size, lanes 0..25 of the wavefront carry one DoF each
constexpr int NQ = WBC_Q_STRIDE;   // 27
constexpr int NL = 32;             // per-lane model tables are padded to 32 entries

// One kinematic model, flattened for "lane j = joint j" and "lane k = velocity column k" access.
// Built by wbc_model_create from WbcModelBlob; lives in device global memory (tiny, L2 resident).
struct DevModel {
  int32_t nq, nv, njoints, maxdepth, nframes, pad_[3];
  // joint lanes
  int32_t parent[NL], depth[NL], jtype[NL], idx_q[NL];
  int32_t ax0[NL], ax1[NL], ax2[NL];   // rotation axis a and its cyclic successors (columns 3*a of the stored R)
  double tp[NL][3];                    // placement translation permuted to (a, a1, a2) order
  double mass[NL], com[NL][3];
  // column lanes
  int32_t col_joint[NL], col_lin[NL], col_ang[NL], col_q[NL];   // local axis index (-1: none); q index of 1-DoF joints
  uint32_t col_subtree[NL];            // joints in the subtree of the column's joint (CoM Jacobian)
  int32_t idx_v_of[NL];                // joint -> its first velocity column
  // controller frames (role order WBC_FR_*)
  int32_t frame_joint[WBC_MAX_FRAMES];
  double frame_p[WBC_MAX_FRAMES][3];
  uint32_t frame_support[WBC_MAX_FRAMES];  // bit k: column k moves the frame
  double total_mass;
};

// Structural presolve plan of one (model, configuration): which stance feet's contact equalities are eliminated and
// the index maps of the reduced problem (built by wbc_batch_configure, read with scalar loads by the tick kernel).
#define WBC_PLAN_NR 16
#define WBC_SIM3_MAXP 16                // constraint rows (original and reduced) the compact sim3 kernel holds
constexpr int WBC_QP_DEFERRED = -1;     // internal status: set by wbc_tick_sim3_kernel, never visible to the caller
struct DevPlan {
  int32_t enabled, nelim, n_red, p_keep;
  int32_t nlock;                     // DoF locked at 0 by the velocity box (>= lock_from, use_bounds): left out of the reduced problem
  int32_t packed_ok;                 // the packed kernel (four instances per wavefront) can run this (model, configuration)
  int32_t packed_ok_pu;              // ... its QCON variant can, given posture_u (and q_con) from the posture kernel or the caller
  int32_t orth;                      // tasks touch the stance legs: contact elimination through an orthonormal null-space basis (contact_presolve_orth)
  int32_t pk_update_ok;              // the packed FK schedule reaches every frame wbc_update_packed_kernel reads, and no trunk task is on
  // packed kernel (wbc_tick_sim3p_kernel): everything a lane needs, one record per role, so that no load depends on another
  struct PkJoint { int32_t joint, parent, a0, a1, a2, rev, q_idx, pad_; double t0, t1, t2; };   // a*: 3 x column of the axis / its successors in R
  struct PkCol { int32_t dof, joint, lin, ang, dq_idx, pad_[3]; double d_lo, d_hi, d_vm; };       // Jacobian column + velocity-damper entries of a DoF
  PkJoint pk_fk[5][16];              // joints of tree depth 2 + L, one per lane-in-instance (joint -1: none)
  PkCol pk_var[16], pk_leg[16];      // reduced variable s / eliminated leg DoF s
  int32_t pk_scq[32];                // joint j (>= 2): q index of its angle if it is a revolute joint the FK needs, else -1
  // packed orth kernel (wbc_tick_orthp_kernel, round 3): the equality-only task problems (BASELINE configs[1]) four instances per wavefront
  int32_t q_ok, q_nred, q_armsup, q_nlock;  // eligible (1: equality-only family; 2: INEQ variant — inequality rows, velocity box, trunk task); n' = 6 + free DoF outside
                                            // base and stance legs (<= 15; INEQ: <= 12, locked DoF left out); bit e: EE task e touches a free (arm) variable; DoF locked at 0
  struct QDof { int32_t joint, lin, ang, bl, red, sub_lo, sub_hi, supmask; };   // DoF d: Jacobian column; position in [base; stance legs] (-1: none);
                                                                                // reduced variable (>= 6, -1: none); joints of its subtree [lo, hi]; bit e: moves EE frame e
  struct QJnt { double m, c0, c1, c2; };    // joint j: mass and centre of mass of its body (joint frame)
  PkJoint q_fk[6][16];               // joints of tree depth 2 + L (all of them: the CoM needs every body)
  int32_t q_scq[32];                 // joint j: q index of its angle (revolute), else -1
  QDof q_dof[32];
  QJnt q_jm[32];
  int32_t q_bl2dof[18], q_red2dof[16];
  int32_t q_efoot[8];                // EE e: index of its leg among the eliminated feet (-1: not an eliminated foot)
  struct XVar { int32_t dof, dq_idx, task, pad_; double d_lo, d_hi, d_vm; };   // DoF, its velocity-damper entries, the EE task that moves with it (-1: none / base)
  XVar q_dmp[32];                    // INEQ variant: the velocity-damper entries of DoF d
  // packed box kernel (wbc_tick_boxp_kernel, round 3): task problems WITHOUT constraint rows (the warm-up problem of setInitialState), four
  // instances per wavefront. The base and (where 16 lanes do not hold the rest) the limb DoF with the widest box are eliminated by a Schur
  // complement; DoF locked at 0 are left out. Shares q_fk / q_scq / q_dof (joint, lin, ang, supmask) with the packed orth kernel.
  int32_t x_ok, x_ne, x_nk, x_nlock;  // eligible; eliminated DoF (6..8), kept (bounded) variables (<= 16), DoF locked at 0
  XVar x_kept[16], x_elim[8];
  int32_t x_role[32];                // DoF d: 0..7 eliminated slot, 16 + k kept variable k, -1 locked / absent
  uint32_t x_limb[16];               // kept variable k: bit k2 = kept variable k2 moves the same task's frame (the limb block of H_KK)
  uint32_t elimrows;                 // bit i: constraint row i belongs to an eliminated foot
  uint32_t legrows;                  // bit i: kept constraint row i has support on eliminated leg DoF (needs C Z)
  // qpJointb "MANI"/"HYBRID" when EVERY finite difference is structurally zero (the perturbed joint is not a proper ancestor
  // of the differentiated joint — all twelve sweeps of the reference's HYBRID indices): no posture kernel is launched, the
  // tick derives u and the leaked state itself.
  int32_t post_static;               // 1: u_i = post_zero bit ? 0 : PREV_i, q_con = q with the post_pert entries at (q+d)-2d
  int32_t post_fk2;                  // 1: an active constraint depends on a perturbed joint -> second FK pass needed
  uint32_t post_zero;                // DoF bits
  uint32_t post_pert;                // q-index bits (literal mode only; 0 when the configuration is restored)
  // qpJointb "MANI" / "HYBRID" with sweeps that DO matter: wbc_posture_par_kernel evaluates every finite-difference point on a lane of its own
  // (the perturbations the reference accumulates are known up front: the configuration of sweep i is q0 with the entries of the earlier
  // sweeps at (q + d) - 2 d). mp_n sweeps (<= 32); sweep k: DoF mp_i, perturbed configuration entry mp_qi, differentiated joint mp_joint, its
  // ancestor chain (joints below the free-flyer, top down, -1 padded), configuration entries perturbed before it (literal mode)
  int32_t mp_ok, mp_n, mp_pad_[2];
  int32_t mp_i[32], mp_qi[32], mp_joint[32];
  int32_t mp_chain[32][8];
  uint32_t mp_prev[32];
  uint32_t mp_all;                   // configuration entries left at (q + d) - 2 d when the loop is through (literal mode)
  uint32_t mp_prevmode;              // DoF bits whose u is the PREV value (HYBRID: the DoF its loop skips); every other DoF: its sweep's value or 0
  uint32_t flags;                    // bit 0 con_com, 1 con_trunk, 2 task_trunk, 3 use_bounds, bits 4..6 task_joint
  uint32_t task_ee_mask, con_ee_mask; // bit e: cfg.task_ee[e] / cfg.con_ee[e] (one scalar instead of five flag loads per loop)
  int32_t rowstart[4];               // first constraint row of eliminated foot f
  int32_t legd[12];                  // DoF index of eliminated leg DoF l (feet in constraint order, DoF ascending)
  int32_t Fd[WBC_PLAN_NR];           // DoF index of reduced variable k (0 beyond n_red)
  uint32_t redsup[WBC_MAX_FRAMES];   // frame f: bit k = reduced variable k moves the frame (frame_support mapped through pos)
  int32_t pos[32];                   // DoF d -> reduced position (-1: eliminated / absent)
  int32_t lidx[32];                  // DoF d -> l (-1: not an eliminated leg DoF)
};

enum Mode : int { MODE_TICK = 0, MODE_ASSEMBLE = 1, MODE_FK = 2 };

struct KernelArgs {
  const DevModel* models;
  const WbcConfig* cfgs;
  const DevPlan* plans;
  int32_t B, mrows, prows, mcart;   // mcart = Cartesian task rows (excludes the diagonal posture block)
  int32_t n_models, pad0_;          // model indices read from the caller's buffer are clamped to [0, n_models)
  int32_t post_static;              // every model's DevPlan.post_static (then no posture kernel runs)
  int32_t dbg_alias;                // diagnostic: every wave loads instance 0's inputs (isolates HBM input latency)
  int32_t dbg_stop;                 // diagnostic (ablation timing): the sim3 kernel returns after stage dbg_stop (0 = run the whole tick)
  int32_t jtj_mfma, presolve;  // presolve: structural elimination of the contact equalities (default on)
  int32_t fk_nj, fk_nf;             // oMi / oMf output strides = max njoints / nframes over the handle's models
  int32_t* defer;                   // [1 + max_batch]: count, then the instances wbc_tick_sim3_kernel left to the general path
  int32_t* pivot_count;             // diagnostic (option "count_pivoted"): instances that took the pivoted elimination; else null
  int32_t* defer_aux;               // [0] the pivot counter's slot, [1] workgroups of the deferred pass that are done, [2] last tick's deferred count
  int32_t dbg_force_defer;          // diagnostic: every instance with a flagged leg block is deferred instead of pivoted
  int32_t packed_trunk;             // packed kernel: the configuration has the trunk task on (the TRUNK variant is launched)
  uint32_t tick_seq;                // packed kernel: this launch's sequence number (the "deferred_last" statistic is (seq, count) in one word)
  unsigned long long* defer_stat;   // packed kernel: that word (instances its tail redid on the general path), or null
  int32_t presolve_orth;            // the orthonormal contact presolve where DevPlan.orth: 0 off, 1 on, 2 on and a plan of this batch has DevPlan.orth (host)
  int32_t orth_qr;                  // diagnostic: the null-space basis always through the Householder QR (else only for flagged leg blocks)
  int32_t refine;                   // iterative-refinement steps at the final working set (QP_Wrapper.py:37 numRefinementSteps; option "refine", default 1)
  // warm start: the final working set of the previous tick, [B][2] words in FULL-problem indexing whatever
  // kernel wrote them: word 0 = velocity bounds (bit d: DoF d at its lower bound, bit 32 + d: at its upper bound), word 1 =
  // constraint rows of findConstraints' order (bit i / 32 + i). Either may be null (cold start / nothing carried); they may alias.
  const unsigned long long* ws_in;
  unsigned long long* ws_out;
  unsigned long long* prof;         // WBC_PROFILE builds: per-phase cycle sums [16] (else unused)
  double dt;
  double sing_tol;                  // a stance-leg block with |det K| <= sing_tol (sum|K_ij|)^3 is not eliminated
  WbcTickIn in;
  WbcTickOut out;
  WbcQpData qp;
  WbcFkOut fk;
};

struct QpArgs {
  int32_t B, n, p, m;               // m > 0: least-squares form (A, b given)
  int32_t dbg_stop;                 // diagnostic (ablation build): the packed QP kernel returns after stage dbg_stop - 400 (0 = the whole solve)
  int32_t use_mfma, refine;         // refine: iterative-refinement steps at the final working set (option "refine", default 1)
  const double *H, *g, *A, *bvec, *C, *lb, *ub, *Clb, *Cub;
  double *x, *H_out, *g_out;
  int32_t *status, *iters;
  const unsigned long long* ws_in;  // hot start (QP.solveQPHotstart): [B][2] working-set words in / out, either may be null, they may alias
  unsigned long long* ws_out;
};

struct IntegrateArgs {
  const DevModel* models;
  int32_t B, n_models;
  double dt;
  const double *q, *v;
  const int32_t* model_id;
  double* q_next;
};

struct PostureArgs {
  const DevModel* models;
  const WbcConfig* cfgs;
  const DevPlan* plans;                // wbc_posture_par_kernel (every model's plan mp_ok); else unused
  int32_t B, n_models;
  const double* q;
  const int32_t* model_id;
  double *u, *q_after;
};

// updateState(running=True) + trunkWorldPos, and (rollout only; every pointer below `q_new` may be null) the reference-state
// side effects of qpb() and the target advance
struct UpdateArgs {
  const DevModel* models;
  const WbcConfig* cfgs;
  const DevPlan* plans;
  int32_t B, n_models;
  int32_t mode, pad_;                  // WBC_ROLLOUT_RUNNING: IMU fed back + trunkWorldPos; WBC_ROLLOUT_WARMUP: q_new = q_next as it is
  const double *q_cur, *q_next, *imu, *foot_targets;
  const int32_t* model_id;
  double* q_new;
  double *ee_target, *prev_ee_target, *trunk_target, *prev_trunk_target, *ee_prev_rot, *trunk_prev_rot;
  const double *ee_ref_rot, *trunk_ref_euler, *ee_step, *trunk_step;
  double* grip_trace;                  // [B][3] of this tick
  const int32_t *status, *iters;       // this tick's
  int32_t *status_max, *iters_sum;
};

// launchers (one per kernel family, wbc_k_*.hip): single-wave workgroups; tick kernels take grid = B, the QP / integrate kernels min(B, resident waves)
int launch_tick(const KernelArgs& a, int mode, int grid, void* stream);
int launch_tick_sim3(const KernelArgs& a, int grid, void* stream);
int launch_tick_sim3p(const KernelArgs& a, void* stream);      // packed: four instances per wavefront, grid = ceil(B / 4)
int launch_tick_boxp(const KernelArgs& a, void* stream);       // packed box kernel (task problems without constraint rows), grid = ceil(B / 4)
int launch_tick_orthp(const KernelArgs& a, void* stream, int ineq);      // packed orth kernel (equality-only task problems), grid = ceil(B / 4)
int orthp_lds_bytes();
int sim3p_lds_bytes();
int launch_tick_deferred(const KernelArgs& a, void* stream);   // general path for the instances the sim3 kernel deferred
int sim3_lds_bytes();
int launch_qp(const QpArgs& a, int grid, void* stream);
int qp_packed_lanes(const QpArgs& a);           // wbc_k_qpp.hip: problems per wavefront the packed QP kernel takes this shape with (4 / 2), 0: not taken
int launch_qp_packed(const QpArgs& a, void* stream);
int launch_integrate(const IntegrateArgs& a, int grid, void* stream);
int launch_posture(const PostureArgs& a, int grid, void* stream);
int launch_posture_par(const PostureArgs& a, int grid, void* stream, int three);   // every finite-difference point on its own lane (DevPlan.mp_ok)
int launch_update(const UpdateArgs& a, int grid, void* stream);
int launch_update_packed(const UpdateArgs& a, void* stream);   // four instances per wavefront (every plan pk_update_ok)
int tick_lds_bytes();

}  // namespace wbc
