/*
 * wbc.h — C-ABI of the MI355X batched whole-body-control hot path.
 *
 * This is the drop-in boundary for the per-tick hot loop of the reference
 * (joey156/MECH5845M-WBC-for-Legged-Manipulator):
 *
 *   RobotModel.runWBC            wrappers/Robot_Wrapper4.py:1330-1412
 *     -> updateState             wrappers/Robot_Wrapper4.py:387-428     (FK, joint Jacobians, frames)
 *     -> qpA / qpb               wrappers/Robot_Wrapper4.py:1271-1294   (task stack A, b)
 *     -> findConstraints         wrappers/Robot_Wrapper4.py:764-836     (C, Clb, Cub)
 *     -> velDamperJointConstraints  wrappers/Robot_Wrapper4.py:572-637  (lb, ub)
 *     -> QP.solveQP / solveQPHotstart  wrappers/QP_Wrapper.py:23-73     (H = A'A, g = -A'b, QP)
 *     -> jointVelocitiestoConfig wrappers/Robot_Wrapper4.py:440-449     (pin.integrate)
 *
 * The reference has no FFI layer (it is pure Python calling pinocchio + qpOASES); the boundary is
 * its two Python classes `QP` and `RobotModel`.  The Python mirrors in
 * mech5845m-wbc-for-legged-manipulator_amd/{QP_Wrapper,Robot_Wrapper4}.py keep those signatures and
 * bind the entry points below through ctypes.
 *
 * Conventions: plain C types only; all matrices row-major fp64; every per-instance array is
 * [B][k] with an instance's k values contiguous; quaternions are (x, y, z, w) as in pinocchio,
 * PyBullet and scipy; Jacobian rows are linear 0-2, angular 3-5 (pinocchio Motion order).
 * Buffers are caller-allocated; `mem` says whether the pointers are host (the library stages them
 * through its own device workspace) or device pointers (used in place, zero copies).
 * With device pointers every call only enqueues kernels (and, on first use of a feature, allocates its workspace):
 * after one warm-up call the same call sequence can be captured into a hipGraph on the caller's stream.
 * Return value: 0 = ok, negative = API-level failure (message via wbc_last_error()).  Per-instance
 * solver outcome goes to the `status` array.  No global state except the thread-local error string;
 * handles are not thread-safe, independent handles may be used concurrently.
 */
#ifndef WBC_H_
#define WBC_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ limits */
#define WBC_MAX_JOINTS 24  /* model joints incl. universe (a1_wx200: 22)                           */
#define WBC_MAX_NQ 28      /* a1_wx200: 27                                                          */
#define WBC_MAX_NV 26      /* a1_wx200: 26 = QP size n; one wavefront lane per DoF   */
#define WBC_NEE 5          /* end effectors in reference order FR, FL, RR, RL, GRIP (sim3.py:56)    */
#define WBC_MAX_FRAMES 16
#define WBC_MAX_P 24       /* constraint rows: CoM 2 + trunk 4 + 5 EE x 3 = 21 max                  */
#define WBC_MAX_M 96       /* task rows accepted by wbc_qp_solve_ls (A is m x n)                    */
#define WBC_MAX_MODELS 4   /* morphologies interleaved in one batch (BASELINE config 5)             */
#define WBC_MFMA_AUTO_ROWS 30 /* "auto" puts H = A'A on the fp64 matrix cores from this many rows of A on    */

/* joint types (pinocchio JointModel*) */
enum { WBC_JT_UNIVERSE = 0, WBC_JT_FF = 1, WBC_JT_RX = 2, WBC_JT_RY = 3, WBC_JT_RZ = 4,
       WBC_JT_PX = 5, WBC_JT_PY = 6, WBC_JT_PZ = 7 };

/* controller frame roles: index into WbcModelBlob.frame_* */
enum { WBC_FR_EE0 = 0 /* ..4: FR, FL, RR, RL foot_fixed, gripper_bar */, WBC_FR_TRUNK = 5,
       WBC_FR_HIP0 = 6 /* ..10: FR/FL/RR/RL hip joint frames, waist */, WBC_FR_ARM_BASE = 11,
       WBC_FR_NROLES = 12 };

enum { WBC_MEM_HOST = 0, WBC_MEM_DEVICE = 1 };

/* per-instance solver status */
enum { WBC_QP_OPTIMAL = 0, WBC_QP_MAX_ITER = 1, WBC_QP_INFEASIBLE = 2, WBC_QP_NUMERICAL = 3 };

/* posture-task mode = RobotModel.task_active_Joint (Robot_Wrapper4.py:1209-1268) */
enum { WBC_JOINT_OFF = 0, WBC_JOINT_TIKHONOV = 1 /* True */, WBC_JOINT_PREV = 2 /* "PREV" */,
       WBC_JOINT_MANI = 3 /* "MANI": manipulability gradient, :1220-1242 */,
       WBC_JOINT_HYBRID = 4 /* "HYBRID": PREV for the quadruped + manipulability gradient for the arm, :1245-1260 */,
       WBC_JOINT_CUSTOM = 5 /* u supplied per instance in WbcTickIn.posture_u */ };

/* API return codes */
enum { WBC_OK = 0, WBC_E_ARG = -1, WBC_E_HIP = -2, WBC_E_UNSUPPORTED = -3, WBC_E_STATE = -4 };

/* ------------------------------------------------------------------ model
 * What pin.buildModelFromUrdf(urdf, JointModelFreeFlyer()) holds that the path reads
 * (Robot_Wrapper4.py:21); produced by tools/bake_model.py, SURVEY.md Appendix A. */
typedef struct WbcModelBlob {
  int32_t nq, nv, njoints;               /* njoints includes universe (index 0)                   */
  int32_t jtype[WBC_MAX_JOINTS];
  int32_t parent[WBC_MAX_JOINTS];
  int32_t idx_q[WBC_MAX_JOINTS];
  int32_t idx_v[WBC_MAX_JOINTS];
  double place_R[WBC_MAX_JOINTS][9];     /* joint placement in parent joint frame                 */
  double place_p[WBC_MAX_JOINTS][3];
  double mass[WBC_MAX_JOINTS];           /* lumped body (fixed children merged), joint frame      */
  double com[WBC_MAX_JOINTS][3];
  double q_lo[WBC_MAX_NQ], q_hi[WBC_MAX_NQ]; /* model.lower/upperPositionLimit (nq-sized)         */
  double v_max[WBC_MAX_NV];                  /* model.velocityLimit (nv-sized)                     */
  int32_t nframes;                       /* >= WBC_FR_NROLES                                      */
  int32_t frame_joint[WBC_MAX_FRAMES];   /* supporting joint                                      */
  double frame_R[WBC_MAX_FRAMES][9];     /* placement in the supporting joint's frame             */
  double frame_p[WBC_MAX_FRAMES][3];
  int32_t ee_joint[WBC_NEE];             /* end_effector_index_list_joint (Robot_Wrapper4.py:49)  */
} WbcModelBlob;

/* ------------------------------------------------------------------ batch-uniform settings
 * = the RobotModel attributes set by __init__/setTasks/setConstraints/staticReachMode
 * (Robot_Wrapper4.py:72-125, 176-193, 1415-1464). 6x6 weights/gains are diagonal in every preset
 * of the reference, so only the diagonals cross the ABI (the Python mirror refuses non-diagonal). */
typedef struct WbcConfig {
  int32_t task_ee[WBC_NEE];    /* task_active_{FR,FL,RR,RL}_foot, task_active_GRIP                */
  int32_t task_trunk;          /* task_active_Trunk                                               */
  int32_t task_com;            /* Robot_Wrapper2 taskActiveCoM (Robot_Wrapper2.py:600-603)        */
  int32_t task_joint;          /* WBC_JOINT_*                                                     */
  int32_t con_com, con_trunk;  /* const_active_CoM / _Trunk                                       */
  int32_t con_ee[WBC_NEE];     /* const_active_{FR,FL,RR,RL}_foot, const_active_GRIP              */
  int32_t use_bounds;          /* 1: velDamperJointConstraints box; 0: no box (equality-only QP)  */
  int32_t lock_from;           /* DoF >= lock_from get lb = ub = 0 (Robot_Wrapper4.py:627-630)    */
  int32_t arm_base_id;         /* model.getJointId(G_base) (Robot_Wrapper4.py:37): HYBRID differentiates joints >= it */
  int32_t posture_literal;     /* MANI/HYBRID: 1 = the reference's arithmetic to the letter (q is perturbed at
                                  the velocity index, perturbations accumulate, and the perturbed configuration is what
                                  findConstraints / velDamperJointConstraints / integrate then see); 0 = the intended
                                  central difference (own q index, configuration restored)                           */
  int32_t damper_qidx[WBC_MAX_NV]; /* which q entry DoF i's damper looks at (quirk C.3)           */
  double damper_lo[WBC_MAX_NV], damper_hi[WBC_MAX_NV], damper_vmax[WBC_MAX_NV];
  double damper_coef, damper_qi, damper_qs;      /* 0.01, 0.026, 0.015 (Robot_Wrapper4.py:574-576) */
  double ee_W[WBC_NEE][6];     /* diag(EE_weight[i])                                              */
  double ee_w[WBC_NEE];        /* cart_task_weight_EE_list[i]                                     */
  double ee_gain[WBC_NEE][6];  /* diag(EE_gains[i]), indexed by EE index as at Robot_Wrapper4.py:908 */
  double trunk_W[6], trunk_w, trunk_gain[6];
  double com_W[3], com_gain[3];
  double joint_w;              /* joint_task_weight                                               */
  double trunk_box_z_frac;     /* 0.25  (Robot_Wrapper4.py:719)                                   */
  double trunk_box_ang;        /* 0.15  (Robot_Wrapper4.py:720-722)                               */
  double trunk_box_scale;      /* 0.5   (Robot_Wrapper4.py:735-736)                               */
  double com_box_scale;        /* 0.8   (Robot_Wrapper4.py:675-677)                               */
} WbcConfig;

/* ------------------------------------------------------------------ per-instance tick inputs
 * (the arguments of runWBC plus the controller state it reads). NULL is allowed where noted. */
typedef struct WbcTickIn {
  const double* q;                 /* [B][WBC_MAX_NQ-1=27] current_joint_config (xyz, quat xyzw, joints) */
  const double* ee_target;         /* [B][5][3] target_cartesian_pos_EE                               */
  const double* prev_ee_target;    /* [B][5][3] prev_EE_pos                                           */
  const double* trunk_target;      /* [B][3]    target_cartesian_pos_trunk     (NULL if no trunk task) */
  const double* prev_trunk_target; /* [B][3]    prev_trunk_ref                 (NULL if no trunk task) */
  const double* trunk_box_center;  /* [B][4]    initial_trunk_pos[2], initial_trunk_ori_euler (NULL if no trunk constraint) */
  const double* ee_ref_rot;        /* [B][5][9] R* = from_euler('xyz', default_EE_ori_list[i]); NULL => R* == R*_prev (omega_ref = 0) */
  const double* ee_prev_rot;       /* [B][5][9] prev_EE_CoM_rot                                       */
  const double* trunk_ref_euler;   /* [B][3]    default_trunk_ori              (NULL if no trunk task) */
  const double* trunk_prev_rot;    /* [B][9]    old_ref_trunk_rot_matrix       (NULL if no trunk task) */
  const double* com_target;        /* [B][3]    Robot_Wrapper2 target_cartesian_pos_CoM (NULL if no CoM task) */
  const double* com_target_vel;    /* [B][3]    Robot_Wrapper2 target_cartesian_vel_CoM               */
  const int32_t* model_id;         /* [B] index into the batch's models; NULL => all model 0          */
  const double* posture_u;         /* [B][26]   posture target u of qpJointb before the (1/nv) w scaling: required for
                                                WBC_JOINT_CUSTOM; for MANI/HYBRID NULL => computed by the library (wbc_posture_target) */
  const double* q_con;             /* [B][27]   configuration seen by findConstraints, velDamperJointConstraints and integrate when it
                                                differs from q (the state qpJointb MANI/HYBRID leaves behind); NULL => q          */
  const uint64_t* working_set;     /* [B][2]    warm start = what solveQPHotstart keeps between ticks (QP_Wrapper.py:55-73, the qpOASES
                                                object's working set): WbcTickOut.working_set of the previous tick. Word 0: velocity
                                                bounds (bit d: DoF d at its lower bound, bit 32 + d: at its upper bound); word 1:
                                                constraint rows in findConstraints' order (bit i: row i at Clb, bit 32 + i: at Cub).
                                                Any bit pattern is accepted: a seed is used only if the equalities-only minimiser
                                                violates it or comes close to it, and wrong seeds are dropped again (restoration
                                                + refresh, csrc/wbc_common.h qp_core); NULL or zeros => cold start */
} WbcTickIn;

#define WBC_Q_STRIDE 27   /* doubles per instance in q / q_next (nq of the largest model)           */
#define WBC_V_STRIDE 26   /* doubles per instance in qdot, g, lb, ub; row length of H, C, A         */

/* the QP data of one tick, as the reference hands it to QP(...) (all optional outputs, [B][...]) */
typedef struct WbcQpData {
  double* A;    /* [B][m][26]  qpA()  (m = wbc_task_rows())          */
  double* b;    /* [B][m]      qpb()                                  */
  double* H;    /* [B][26][26] A'A   (QP_Wrapper.py:17)               */
  double* g;    /* [B][26]     -A'b  (QP_Wrapper.py:18)               */
  double* C;    /* [B][p][26]  findConstraints() before its final .T  */
  double* Clb;  /* [B][p]                                             */
  double* Cub;  /* [B][p]                                             */
  double* lb;   /* [B][26]     velDamperJointConstraints()            */
  double* ub;   /* [B][26]                                            */
} WbcQpData;

typedef struct WbcTickOut {
  double* qdot;     /* [B][26] xOpt; all zeros for an instance whose status is not WBC_QP_OPTIMAL (the reference's xOpt on
                       its first QP: qpOASES does not write the vector of an unsolved problem, QP_Wrapper.py:50-51) */
  int32_t* status;  /* [B] WBC_QP_*                                                       */
  int32_t* iters;   /* [B] working-set changes (may be NULL)                              */
  double* q_next;   /* [B][27] pin.integrate(q, qdot*dt) (may be NULL)                    */
  uint64_t* working_set; /* [B][2] the inequality constraints active at the solution, format of WbcTickIn.working_set (may be
                            NULL; may alias the input); zeros for an instance whose QP was not solved */
} WbcTickOut;

/* forward-kinematics outputs of updateState (all optional). In a handle with several models the per-instance strides of
 * oMi / oMf are the LARGEST model's njoints / nframes; an instance of a smaller model fills its own rows and zeroes the rest. */
typedef struct WbcFkOut {
  double* oMi;     /* [B][max njoints][12]  R (9, row-major) then p (3); joint 0 = identity   */
  double* oMf;     /* [B][max nframes][12]  controller frames (WbcModelBlob.frame_*)          */
  double* J;       /* [B][6][26] data.J of computeJointJacobians (WORLD)                      */
  double* com;     /* [B][3]     data.com[0]                                                  */
  double* Jcom;    /* [B][3][26] jacobianCenterOfMass                                         */
} WbcFkOut;

typedef struct WbcModel WbcModel;
typedef struct WbcBatch WbcBatch;

/* ------------------------------------------------------------------ entry points */

/* replaces pin.buildModelFromUrdf + createData (Robot_Wrapper4.py:21-23); validates the blob. */
int wbc_model_create(const WbcModelBlob* blob, WbcModel** out);
void wbc_model_destroy(WbcModel* m);

/* one handle per device/stream: device workspace for up to max_batch instances.
 * n_models == 0 gives a QP-only handle (wbc_qp_solve / wbc_qp_solve_ls), as QP_Wrapper.QP needs no robot model. */
int wbc_batch_create(const WbcModel* const* models, int n_models, int max_batch, int device_id, WbcBatch** out);
void wbc_batch_destroy(WbcBatch* b);

/* replaces setTasks / setConstraints / staticReachMode and the weight attributes
 * (Robot_Wrapper4.py:176-193, 1415-1464); cfg applies to model `model_index` of the batch. */
int wbc_batch_configure(WbcBatch* b, int model_index, const WbcConfig* cfg);
int wbc_task_rows(const WbcBatch* b);       /* m of qpA() under the current switches        */
int wbc_constraint_rows(const WbcBatch* b); /* p of findConstraints()                        */

/* replaces the pinocchio calls of updateState (Robot_Wrapper4.py:400-405) and
 * pin.jacobianCenterOfMass (Robot_Wrapper4.py:670). */
int wbc_fk_jacobians(WbcBatch* b, int B, const double* q, const int32_t* model_id, int mem,
                     const WbcFkOut* out, void* stream);

/* replaces qpA/qpb/findConstraints/velDamperJointConstraints + QP.__init__'s H, g
 * (Robot_Wrapper4.py:1348-1361, QP_Wrapper.py:17-18). */
int wbc_assemble(WbcBatch* b, int B, const WbcTickIn* in, double dt, int mem, const WbcQpData* out, void* stream);

/* replaces QP.solveQP / solveQPHotstart given H, g (QP_Wrapper.py:23-73):
 * argmin 1/2 x'Hx + g'x  s.t. lb <= x <= ub, Clb <= C x <= Cub.
 * n <= 26, p <= WBC_MAX_P; H is [B][n][n], C is [B][p][n] (row-major, NOT the reference's C.T view —
 * SURVEY.md C.1); lb/ub/C may be NULL (no box / no rows).
 * Hot start (solveQPHotstart, QP_Wrapper.py:55-73: qpOASES keeps its working set between calls): working_set_in / working_set_out,
 * [B][2] words in the QP's own indexing — word 0: bit i = variable i at its lower bound, bit 32 + i = at its upper bound; word 1: the
 * same for constraint row i. Either may be NULL (cold start / nothing returned); they may be the same buffer. A seed is only a hint:
 * the answer is the cold solve's (H > 0), wrong seeds are dropped again. An unsolved QP returns an empty set.
 * Kernel: several problems per wavefront (csrc/wbc_k_qpp.hip: four when n <= 16 and p <= 16, else two; statistic "last_qp_path" = 4 / 2),
 * cold and hot start alike; option "packed_kernel" 0 keeps one problem per wavefront (csrc/wbc_k_misc.hip, "last_qp_path" = 1). Same status,
 * iteration count and working set from both; answers equal to rounding. */
int wbc_qp_solve(WbcBatch* b, int B, int n, int p, const double* H, const double* g, const double* C,
                 const double* lb, const double* ub, const double* Clb, const double* Cub, int mem,
                 double* x, int32_t* status, int32_t* iters, const uint64_t* working_set_in, uint64_t* working_set_out,
                 void* stream);

/* replaces QP(A, b, ...) + solveQP(): forms H = A'A, g = -A'b on the device (QP_Wrapper.py:17-18)
 * and solves. A is [B][m][n], m <= WBC_MAX_M. H_out/g_out optional. The packed kernel (see wbc_qp_solve) forms A'A on the fp64 matrix cores
 * always; use_mfma chooses inside the one-per-wavefront kernel only: 1 = fp64 MFMA contraction, 0 = VALU, -1 = MFMA from WBC_MFMA_AUTO_ROWS rows on.
 * With option "refine" (default 1) the answer gets one step of iterative refinement from A, b themselves where the problem's pivot ratio asks
 * for it (csrc/wbc_common.h WBC_REFINE_COND; QP_Wrapper.py:37 numRefinementSteps); wbc_qp_solve (H, g alone) is never refined.
 * working_set_in / working_set_out: as in wbc_qp_solve (QP.solveQPHotstart passes the previous call's set). */
int wbc_qp_solve_ls(WbcBatch* b, int B, int m, int n, int p, const double* A, const double* bvec, const double* C,
                    const double* lb, const double* ub, const double* Clb, const double* Cub, int mem, int use_mfma,
                    double* x, int32_t* status, int32_t* iters, double* H_out, double* g_out,
                    const uint64_t* working_set_in, uint64_t* working_set_out, void* stream);

/* replaces qpJointb's "MANI" / "HYBRID" branches (Robot_Wrapper4.py:1220-1260) under the configured task_joint,
 * arm_base_id and posture_literal: u [B][26] = the posture target before scaling (PREV / zeros for the other modes),
 * q_after [B][27] = the configuration the reference's state holds afterwards (optional). wbc_tick / wbc_assemble call
 * this themselves when task_joint is MANI or HYBRID and WbcTickIn.posture_u is NULL. */
int wbc_posture_target(WbcBatch* b, int B, const double* q, const int32_t* model_id, int mem, double* u, double* q_after,
                       void* stream);

/* the fused hot path = one runWBC tick up to and including the QP (+ optional integrate):
 * FK -> Jacobians -> task stack -> H, g, C, bounds -> QP -> qdot [-> q_next]. */
int wbc_tick(WbcBatch* b, int B, const WbcTickIn* in, double dt, int mem, const WbcTickOut* out, void* stream);

/* replaces jointVelocitiestoConfig / pin.integrate (Robot_Wrapper4.py:440-441): q_next = q (+) v*dt */
int wbc_integrate(WbcBatch* b, int B, const double* q, const double* v, const int32_t* model_id, double dt, int mem,
                  double* q_next, void* stream);

/* replaces the tail of runWBC, updateState(joint_config, base_config, running=True) (Robot_Wrapper4.py:1397-1399, 387-428)
 * with the foot-anchored base estimator trunkWorldPos (:1297-1327):
 *   q_new = [base xyz re-estimated from the stance-foot targets, imu quaternion, joints of q_next].
 * q_cur: current_joint_config (its base xyz is read); q_next: wbc_tick's q_next; imu [B][4] (x, y, z, w) = base_config,
 * NULL => the quaternion of q_next; foot_targets [B][5][3] = the tick's ee_target (FR, FL, RR, RL are read).
 * q_new may alias q_cur. */
int wbc_update_state(WbcBatch* b, int B, const double* q_cur, const double* q_next, const double* imu,
                     const double* foot_targets, const int32_t* model_id, int mem, double* q_new, void* stream);

/* K closed-loop ticks without leaving the device (the 2000 QPs of setInitialState for B robots in one
 * call with mode = WBC_ROLLOUT_WARMUP, ticks = hold_ticks = 1000): per tick wbc_tick -> wbc_update_state -> the
 * reference-state side effects of qpb() (prev_EE_pos / prev_EE_CoM_rot, calcTargetVelEE3 :1151-1152; prev_trunk_ref /
 * old_ref_trunk_rot_matrix, calcTargetVelTrunk2 :995-996) -> the targets advance by a per-tick step (one linear segment
 * of sim3.py's milestone trajectory, sim3.py:207-228). `in0` is the state and the targets of the first tick (never
 * written); all outputs optional. */
enum { WBC_ROLLOUT_RUNNING = 0  /* updateState(joint_config, base_config, running=True): IMU quaternion fed back, base xyz
                                   re-estimated from the stance-foot targets (trunkWorldPos)                            */,
       WBC_ROLLOUT_WARMUP = 1   /* updateState(new_config, feedback=False, running=False) as inside setInitialState's loop
                                   (Robot_Wrapper4.py:287-326, 440-447 with initialised == False): the integrated
                                   configuration becomes the state as it is — free-floating base, no IMU, no estimator   */ };
typedef struct WbcRollout {
  int32_t ticks;                    /* K >= 1 ticks during which the targets advance by their steps        */
  int32_t mode;                     /* WBC_ROLLOUT_*                                                        */
  const double* ee_target_step;     /* [B][5][3] added to ee_target after every tick; NULL => constant     */
  const double* trunk_target_step;  /* [B][3]; NULL => constant                                            */
  const double* imu;                /* [B][4] base quaternion fed back every tick; NULL => the integrated one */
  double* q_final;                  /* [B][27] current_joint_config after K ticks                          */
  double* qdot_last;                /* [B][26] xOpt of the last tick                                       */
  double* ee_target_final;          /* [B][5][3] targets the NEXT tick would get                           */
  double* grip_trace;               /* [K][B][3] gripper_bar position reached after every tick (sim3.py:340-348's log) */
  int32_t* status_max;              /* [B] worst WBC_QP_* status over the K ticks                          */
  int32_t* iters_sum;               /* [B] working-set changes over the K ticks                            */
  int32_t hold_ticks;               /* further ticks with the targets held where the K ticks left them (the second, clamped
                                       segment of setInitialState's trajectories, Robot_Wrapper4.py:275); grip_trace then holds
                                       ticks + hold_ticks entries                                           */
  int32_t pad_;
} WbcRollout;
int wbc_rollout(WbcBatch* b, int B, const WbcTickIn* in0, double dt, const WbcRollout* r, int mem, void* stream);

/* Knobs of a handle (none of them changes a result beyond rounding; defaults in brackets):
 *   "jtj_mfma"        [-1] H = A'A of wbc_tick / wbc_assemble (QP_Wrapper.py:17) on the fp64 matrix cores (v_mfma_f64_16x16x4_f64)
 *                          or as the sparse vector-unit contraction. -1: matrix cores when the Cartesian task stack has
 *                          >= WBC_MFMA_AUTO_ROWS rows (measured: +9 % ticks/s at 33 and 45 rows, a wash at 6); 0: never;
 *                          1: always — that also selects the general tick kernel (the compact sim3 kernel has no matrix-core
 *                          path), so on the sim3 switch set it costs the structural speed-up.
 *   "presolve"         [1] structural elimination of the stance-foot contact equalities (Robot_Wrapper4.py:757-761) where no
 *                          task touches the stance legs; 0: every QP runs at its full size n = nv.
 *   "presolve_orth"    [1] the same elimination where tasks DO touch the stance legs (foot / trunk / CoM tasks, e.g. BASELINE
 *                          configs[1]), through an orthonormal basis of the contact rows' null space (Householder QR per instance):
 *                          the reduced QP is as well conditioned as the full one; the base and stance-leg velocity bounds become
 *                          6 + 3 x (stance feet) two-sided rows. 0: those configurations run at full size. Needs "presolve".
 *   "orth_qr"          [0] diagnostic: that basis always by the Householder QR of the contact rows; by default only instances with a
 *                          nearly rank-deficient stance-leg block take it, the others orthonormalise [I; -K^-1 B] through a 6 x 6
 *                          Cholesky factor (same null space, a quarter of the instructions).
 *   "sim3_kernel"      [1] batches whose every model has such an elimination plan, <= 16 constraint rows and no orientation
 *                          references run on the compact wbc_tick_sim3_kernel (+ a second pass of the general kernel over the
 *                          instances whose leg blocks it could not eliminate); 0: the general kernel does the elimination.
 *   "packed_kernel"    [1] batches of the sim3 switch-set family itself (Grip task or none, optionally the trunk task, trunk box + foot
 *                          contacts, velocity bounds, posture PREV / Tikhonov / static HYBRID — or any mode with posture_u / q_con
 *                          supplied by the posture kernel or the caller: the QCON variant; the gripper's orientation reference is
 *                          honoured; working sets in and out are taken: the WARM variant) run FOUR instances per wavefront
 *                          (wbc_tick_sim3p_kernel: ONE kernel per tick — an instance it cannot reduce, a stance-leg block of rank < 2, is
 *                          redone by its own wave on the general path at the end of the same kernel); 0: one instance per wavefront
 *                          (wbc_tick_sim3_kernel + second pass). Also gates "packed_orth", "packed_box" and the packed QP kernel of
 *                          wbc_qp_solve / wbc_qp_solve_ls (0: one problem per wavefront there too).
 *   "packed_orth"      [1] the equality-only task problems (BASELINE configs[1]: EE tasks + CoM task + posture Tikhonov / PREV, foot
 *                          contacts the only constraints, no velocity box) run FOUR instances per wavefront on wbc_tick_orthp_kernel
 *                          (orthonormal contact presolve, unconstrained reduced problem) from 4608 instances on — below that one round
 *                          of waves covers the batch and the one-instance kernel's shorter dependent chain wins (25 vs 55 us at
 *                          B = 1024, equal at 4096, 0.29 vs 0.78 ms at 65536); 2: at every batch size; 0: never. Its INEQ variant covers
 *                          the same task problems WITH inequality rows (trunk box, CoM box, the velocity box — rows of Z in the reduced
 *                          coordinates, two per lane — and the trunk task: tests' "everything", 0.58 vs 1.24 ms at 65536), cold or hot-started.
 *   "packed_box"       [1] the task problems WITHOUT constraint rows (the warm-up problem of setInitialState, Robot_Wrapper4.py:196-351:
 *                          trunk / EE tasks + posture Tikhonov / PREV, velocity box only) run FOUR instances per wavefront on
 *                          wbc_tick_boxp_kernel: the base and, where 16 lanes do not hold the rest, the limb DoF with the widest box are
 *                          eliminated by a Schur complement, the dual method works on <= 16 bounded unknowns, and an instance whose
 *                          optimum holds an eliminated DoF at its own velocity bound (or more than 12 active bounds) is redone by
 *                          its wave on the general path. Working sets in / out: its WARM variant. At every batch size (28 vs 29 us at
 *                          B = 1, 40 vs 46 us at 1024, 0.37 vs 1.23 ms at 65536); 0: never (general kernel).
 *   "posture_par"      [1] qpJointb "MANI" / "HYBRID" targets on the parallel posture kernels: every sweep on a lane of its own (both sides of
 *                          the central difference one after the other), three instances per wavefront, where no model has more than 21
 *                          sweeps (statistic "last_posture_par" = 2), else every finite-difference point on a lane of its own, one instance
 *                          per wavefront (= 1; value 3 forces this form); 0: the sequential whole-tree kernel (52 sweeps per instance).
 *   "packed_update"    [1] wbc_update_state / the roll-out's state update run four instances per wavefront where every model's
 *                          configuration is of the packed kernel's family (statistic "last_update_packed"); 0: one per wavefront.
 *   "presolve_tol_exp" [7] a stance-leg 3 x 3 block K with |det K| <= 10^-value (sum |K_ij|)^3 is treated as rank deficient: the
 *                          compact kernel eliminates it with column pivoting and keeps one leg velocity + one contact equality
 *                          in the reduced QP (the general kernel's in-kernel presolve falls back to the full problem).
 *                          0: every block of every instance takes that path.
 *   "count_pivoted"    [0] diagnostic: count the instances that took the pivoted elimination (statistic "pivoted_last"; one
 *                          atomic per such instance — leave it off when timing).
 *   "dbg_force_defer"  [0] diagnostic: an instance with a flagged block takes the path of a block of rank < 2 instead of the pivoted
 *                          elimination — the packed kernel's tail (general path, same kernel), the one-instance kernel's second pass.
 *   "warm_start"       [0] 1: wbc_rollout carries every instance's final working set into its next tick (the hot start the
 *                          reference gets from QP.solveQPHotstart, Robot_Wrapper4.py:1389-1394); 0: every tick starts cold.
 *                          Same minimiser either way (H > 0). Warm ticks stay on the packed kernel (its WARM variant: seeds through
 *                          the add step, x / u rebuilt from the factors, restoration). Measured on MI355X at B = 65536
 *                          (profiles/r03_rollout_warm_vs_cold.txt): closed loop 277.5 vs 280.6 M ticks/s un-stressed (-1 %), 190.0
 *                          vs 197.7 M on the stress recipe (-4 %); open-loop ticks seeded with their own set +2 % / 0 %, with the
 *                          previous tick's -1.5 %: a dual method that starts at the unconstrained minimiser needs one iteration
 *                          per active inequality (0.9-1.05 here), a seed replaces it by an add step + a share of the rebuild, and
 *                          both pay the final feasibility scan. Off by default; never the 2.2x of round 2's fallback kernel.
 *                          (wbc_tick is warm exactly when WbcTickIn.working_set is passed.)
 *   "refine"           [1] one step of iterative refinement at the final working set (0 = off) — the analogue of the reference's
 *                          qpOASES option numRefinementSteps = 100 (QP_Wrapper.py:37). With the active normals N, their multipliers u
 *                          and the dual method's factors (J J' = H^-1, J'N' = [R; 0]):  r1 = -(grad f - N'u),  x += J2 J2' r1 (+ the
 *                          active rows' own residual through R). grad f is formed from the UNFACTORED least-squares data — A'(A x - b)
 *                          as two products, task block by task block, never through H = A'A: fl(A'A) carries the benchmark tick's
 *                          1.5e-9 posture block with 1e-5 relative error, which is the whole 1e-6 the plain method is off by at
 *                          cond(H) ~ 3e9. One step lands within 1e-8 of the exact least-squares optimum (oracle: qp_refine; both sides
 *                          refine, so the parity margin on BASELINE configs[2] went from 5.7e-6 to 6.6e-8 against the 1e-5 tolerance,
 *                          profiles/r04_soak_long.txt). A correction larger than 0.25 max(1, |x|) or non-finite is not applied; on wbc_qp_solve_ls (arbitrary
 *                          problems) a QP whose smallest Cholesky pivot is above 1e-5 x its largest diagonal entry skips the step — the ticks refine always. Cost on the benchmark: 5.5 % (profiles/r04_ab_refine.txt). Refined: the packed
 *                          sim3 kernel and its variants, the general kernel (full size and structural presolve), wbc_qp_solve_ls
 *                          (wbc_qp_solve, given H and g alone, could only use -(H x + g): no gain where H itself is the rounding — measured — so it does not refine). Not refined, because their
 *                          stacks are well conditioned (1e-8 .. 1e-10 without): the packed orth / box kernels and the orthonormal
 *                          presolve; and the one-instance compact kernel (path 1: no room at 168 VGPRs / 13 KB LDS) — with refine > 0
 *                          what it would take runs on the general kernel instead (refine = 0 brings it back).
 *   "packed_min_batch" [1] the packed sim3 kernel takes batches from this many instances on. 1: it is the fastest path of the sim3
 *                          family at EVERY batch size (13.6 us at B = 1 against 19 us on either one-instance kernel, 41 vs 53-55 us at
 *                          1024: profiles/r04_small_batch_c3.txt).
 *   "grid"                 workgroups of the grid-stride kernels (wbc_qp_solve*, wbc_integrate); default = what fills the chip.
 *   "dbg_alias_inputs" [0] diagnostic: every instance reads instance 0's inputs (isolates input latency in timings).
 *   "dbg_stop"         [0] diagnostic (ablation build of the library only): the sim3 kernel stops after stage k (1..7, see csrc/wbc_k_sim3.hip;
 *                          101.. the packed sim3 kernel, 201.. the packed orth kernel, 301.. the packed box kernel) — outputs are garbage,
 *                          only the run time means something (tools/ablate_sim3.py, ablate_orthp.py, ablate_boxp.py). */
int wbc_batch_set_option(WbcBatch* b, const char* name, int value);

/* Read-only statistics of a handle: "last_path" (kernel the last tick ran on: 0 general, 1 compact sim3 + second pass, 2 packed
 * compact sim3 — four instances per wavefront, one kernel —, 3 packed orth kernel, 4 packed box kernel), "last_orth" (1: that tick ran with the
 * orthonormal contact presolve, option "presolve_orth": the general kernel's ORTH variant or the packed orth kernel),
 * "last_qp_path" (problems per wavefront of the last wbc_qp_solve / wbc_qp_solve_ls call: 4, 2 or 1),
 * "last_update_packed" (1: the last state update ran on the packed kernel), "last_posture_par" (1 / 2: the last MANI / HYBRID posture target
 * ran on the parallel posture kernel, one / three instances per wavefront), "deferred_last" (instances the last tick's kernel could not reduce itself: redone in the packed kernels'
 * tail or left to the one-instance kernel's second pass; waits for `stream`), "pivoted_last" (instances that took the pivoted
 * elimination, with option "count_pivoted"), "sim3_lds_bytes" / "tick_lds_bytes" / "orthp_lds_bytes" (LDS per workgroup of the tick kernels). */
int wbc_batch_get_stat(WbcBatch* b, const char* name, void* stream, int64_t* out);

/* wait for everything queued by this handle on `stream`. */
int wbc_batch_synchronize(WbcBatch* b, void* stream);

/* diagnostics: per-phase shader-cycle sums of wbc_tick since the last call, filled only by the -DWBC_PROFILE build of
 * the library (zeros otherwise): [0] ticks, [1] FK+Jacobians, [2] task stack, [3] Cholesky, [4] L^-1 and x0,
 * [5] equality phase, [6] inequality phase, [7] output/integrate, [8] working-set changes, [9..12] task-stack sub-phases, [13..19] contact-presolve sub-phases (24 values). */
int wbc_debug_cycles(WbcBatch* b, uint64_t* out24);

const char* wbc_last_error(void);
const char* wbc_version(void);
int wbc_abi_sizes(int32_t* sizeof_blob, int32_t* sizeof_config); /* ctypes layout self-check */

#ifdef __cplusplus
}
#endif
#endif /* WBC_H_ */
