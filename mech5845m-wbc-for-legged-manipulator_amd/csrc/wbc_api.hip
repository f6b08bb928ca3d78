// wbc_api.hip — the C-ABI of include/wbc.h on top of the gfx950 kernels (handles, validation, host staging).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <new>
#include <vector>
#include "wbc_device.h"

using namespace wbc;

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return fail(WBC_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));   \
  } while (0)

struct WbcModel {
  WbcModelBlob blob;
  DevModel dev;
};

// The packed sim3 kernel's batch-size policy (tools/small_batch.py, profiles/r04_small_batch_c3.txt): below this many instances a launch is a
// single partial round of waves and one tick's latency is what counts.
constexpr int WBC_SIM3P_MIN_BATCH = 1;

struct WbcBatch {
  int device_id, n_models, max_batch, grid;
  const WbcModel* models[WBC_MAX_MODELS];
  WbcConfig cfg_host[WBC_MAX_MODELS];
  DevPlan plan_host[WBC_MAX_MODELS];
  bool configured[WBC_MAX_MODELS];
  DevModel* d_models;
  WbcConfig* d_cfgs;
  DevPlan* d_plans;
  void* ws;
  size_t ws_bytes;
  int mrows, prows, mcart;
  int jtj_mfma;
  int presolve, presolve_orth, orth_qr;
  double sing_tol;
  int sim3_kernel;       // 1 (default): batches that qualify run on wbc_tick_sim3_kernel (compact LDS) + a deferred pass
  int32_t* d_status;     // status buffer of our own when the caller passes none (the deferred pass needs one)
  int dbg_alias, dbg_stop;
  int count_pivoted, force_defer;   // diagnostics of the sim3 kernel's pivoted elimination / second pass
  int packed_min_batch;  // the packed sim3 kernel takes batches from this many instances on (default WBC_SIM3P_MIN_BATCH; option "packed_min_batch")
  int packed_kernel;     // 1 (default): eligible batches run four instances per wavefront (wbc_tick_sim3p_kernel)
  int posture_par, last_posture_par;   // option [1]: MANI / HYBRID posture targets on wbc_posture_par_kernel (every finite-difference point on its own lane); what the last one ran on
  int packed_box;        // 1 (default): task problems without constraint rows (the warm-up problem) run four instances per wavefront (wbc_tick_boxp_kernel)
  int packed_orth;       // 1 (default): equality-only task problems run four instances per wavefront (wbc_tick_orthp_kernel)
  int refine;            // iterative-refinement steps at the final working set (default 1; 0 = the plain dual method: tests / A-B timing)
  int warm_start;        // 1: wbc_rollout carries each instance's working set from tick to tick (default 0: measured slower)
  int32_t* d_defer;      // [1 + max_batch]: count + compact list of the instances the sim3 kernel deferred (lazy)
  unsigned long long* d_dstat;   // packed kernel: (launch sequence, instances its tail redid on the general path) (lazy)
  uint32_t tick_seq;
  int packed_update, last_update_packed;   // option: wbc_update_packed_kernel where every plan allows it [1]; what the last update ran on
  int last_orth;         // the last general-kernel tick ran the variant with the orthonormal contact presolve
  int last_qp_path;      // problems per wavefront of the last wbc_qp_solve / wbc_qp_solve_ls: 1 (wbc_qp_kernel), 2 or 4 (wbc_qp_packed_kernel)
  int last_path;         // kernel the last wbc_tick / wbc_rollout tick ran on: 0 general, 1 sim3 (+ deferred pass), 2 packed sim3, 3 packed orth, 4 packed box
  int max_nj, max_nf;    // FK output strides: the largest model's joint / frame counts
  unsigned long long* d_prof;
  double *d_pu, *d_pq;   // qpJointb MANI/HYBRID results: u [max_batch][26], q_after [max_batch][27] (lazy)
  void* d_roll;          // wbc_rollout's mutable controller state for max_batch instances (lazy)
};

// ---------------------------------------------------------------------------------------------- model
static bool is_identity(const double* R) {
  static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < 9; ++i)
    if (R[i] != I[i]) return false;
  return true;
}

extern "C" int wbc_model_create(const WbcModelBlob* b, WbcModel** out) {
  if (!b || !out) return fail(WBC_E_ARG, "wbc_model_create: null argument");
  if (b->njoints < 2 || b->njoints > WBC_MAX_JOINTS || b->nv < 6 || b->nv > WBC_MAX_NV || b->nq != b->nv + 1 || b->nq > WBC_Q_STRIDE)
    return fail(WBC_E_ARG, "wbc_model_create: sizes out of range (njoints %d, nq %d, nv %d)", b->njoints, b->nq, b->nv);
  if (b->jtype[1] != WBC_JT_FF || b->parent[1] != 0 || b->idx_q[1] != 0 || b->idx_v[1] != 0)
    return fail(WBC_E_UNSUPPORTED, "wbc_model_create: joint 1 must be the free-flyer root (Robot_Wrapper4.py:21)");
  if (b->nframes < WBC_FR_NROLES || b->nframes > WBC_MAX_FRAMES) return fail(WBC_E_ARG, "wbc_model_create: nframes %d", b->nframes);
  WbcModel* m = new (std::nothrow) WbcModel;
  if (!m) return fail(WBC_E_ARG, "out of memory");
  m->blob = *b;
  DevModel& d = m->dev;
  memset(&d, 0, sizeof d);
  d.nq = b->nq; d.nv = b->nv; d.njoints = b->njoints; d.nframes = b->nframes;
  int depth[WBC_MAX_JOINTS] = {0};
  double total = 0;
  for (int j = 1; j < b->njoints; ++j) {
    const int t = b->jtype[j];
    if (b->parent[j] < 0 || b->parent[j] >= j) { delete m; return fail(WBC_E_ARG, "joint %d: parent %d not before it", j, b->parent[j]); }
    if (j > 1 && !(t >= WBC_JT_RX && t <= WBC_JT_PZ)) { delete m; return fail(WBC_E_UNSUPPORTED, "joint %d: type %d unsupported on device", j, t); }
    if (!is_identity(b->place_R[j])) { delete m; return fail(WBC_E_UNSUPPORTED, "joint %d: rotated joint placement unsupported on device", j); }
    if (j > 1 && b->parent[j] < 1) { delete m; return fail(WBC_E_UNSUPPORTED, "joint %d: only one root joint supported", j); }
    if (j == 1 && (b->place_p[1][0] != 0 || b->place_p[1][1] != 0 || b->place_p[1][2] != 0)) { delete m; return fail(WBC_E_UNSUPPORTED, "root joint placement must be identity"); }
    depth[j] = depth[b->parent[j]] + 1;
    if (depth[j] > d.maxdepth) d.maxdepth = depth[j];
    d.parent[j] = b->parent[j]; d.depth[j] = depth[j]; d.jtype[j] = t; d.idx_q[j] = b->idx_q[j]; d.idx_v_of[j] = b->idx_v[j];
    const int a = (t >= WBC_JT_RX && t <= WBC_JT_RZ) ? t - WBC_JT_RX : (t >= WBC_JT_PX ? t - WBC_JT_PX : 0);
    d.ax0[j] = a; d.ax1[j] = (a + 1) % 3; d.ax2[j] = (a + 2) % 3;
    d.tp[j][0] = b->place_p[j][d.ax0[j]]; d.tp[j][1] = b->place_p[j][d.ax1[j]]; d.tp[j][2] = b->place_p[j][d.ax2[j]];
    d.mass[j] = b->mass[j];
    for (int i = 0; i < 3; ++i) d.com[j][i] = b->com[j][i];
    total += b->mass[j];
    // velocity columns of this joint
    const int nvj = (t == WBC_JT_FF) ? 6 : 1;
    if (b->idx_v[j] < 0 || b->idx_v[j] + nvj > b->nv) { delete m; return fail(WBC_E_ARG, "joint %d: idx_v out of range", j); }
    // idx_q becomes a global write index (q_next) and a shift count in the kernels: 1-DoF joints live in [7, nq)
    if (t != WBC_JT_FF && (b->idx_q[j] < 7 || b->idx_q[j] >= b->nq)) { delete m; return fail(WBC_E_ARG, "joint %d: idx_q %d outside [7, nq = %d)", j, b->idx_q[j], b->nq); }
    for (int c = 0; c < nvj; ++c) {
      const int k = b->idx_v[j] + c;
      d.col_joint[k] = j; d.col_lin[k] = -1; d.col_ang[k] = -1; d.col_q[k] = b->idx_q[j] + c;
      if (t == WBC_JT_FF) { if (c < 3) d.col_lin[k] = c; else d.col_ang[k] = c - 3; }
      else if (t <= WBC_JT_RZ) d.col_ang[k] = a;
      else d.col_lin[k] = a;
    }
  }
  d.total_mass = total;
  auto on_path = [&](int anc, int j) { while (j > 0) { if (j == anc) return true; j = b->parent[j]; } return false; };
  for (int k = 0; k < b->nv; ++k) {
    uint32_t mask = 0;
    for (int j = 1; j < b->njoints; ++j) if (on_path(d.col_joint[k], j)) mask |= 1u << j;
    d.col_subtree[k] = mask;
  }
  for (int f = 0; f < b->nframes; ++f) {
    const int jf = b->frame_joint[f];
    if (jf < 1 || jf >= b->njoints) { delete m; return fail(WBC_E_ARG, "frame %d: supporting joint %d out of range", f, jf); }
    if (!is_identity(b->frame_R[f])) { delete m; return fail(WBC_E_UNSUPPORTED, "frame %d: rotated frame offset unsupported on device", f); }
    d.frame_joint[f] = jf;
    for (int i = 0; i < 3; ++i) d.frame_p[f][i] = b->frame_p[f][i];
    uint32_t mask = 0;
    for (int k = 0; k < b->nv; ++k) if (on_path(d.col_joint[k], jf)) mask |= 1u << k;
    d.frame_support[f] = mask;
  }
  *out = m;
  return WBC_OK;
}

extern "C" void wbc_model_destroy(WbcModel* m) { delete m; }

// ---------------------------------------------------------------------------------------------- batch
extern "C" void wbc_batch_destroy(WbcBatch* b);
extern "C" int wbc_batch_create(const WbcModel* const* models, int n_models, int max_batch, int device_id, WbcBatch** out) {
  if (!out || n_models < 0 || n_models > WBC_MAX_MODELS || (n_models > 0 && !models) || max_batch < 1)
    return fail(WBC_E_ARG, "wbc_batch_create: bad arguments (n_models %d, max_batch %d)", n_models, max_batch);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(WBC_E_HIP, "wbc_batch_create: no HIP device visible (this library has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return fail(WBC_E_ARG, "wbc_batch_create: device %d of %d", device_id, ndev);
  HIP_TRY(hipSetDevice(device_id));
  WbcBatch* b = new (std::nothrow) WbcBatch;
  if (!b) return fail(WBC_E_ARG, "out of memory");
  memset(b, 0, sizeof *b);
  b->device_id = device_id; b->n_models = n_models; b->max_batch = max_batch; b->presolve = 1; b->presolve_orth = 1; b->packed_update = 1; b->sim3_kernel = 1; b->sing_tol = 1e-7;
  b->jtj_mfma = -1; b->refine = 1; b->packed_min_batch = WBC_SIM3P_MIN_BATCH; b->warm_start = 0; b->packed_kernel = 1; b->packed_orth = 1; b->packed_box = 1; b->posture_par = 1;
  std::vector<DevModel> dm(n_models);
  for (int i = 0; i < n_models; ++i) {
    if (!models[i]) { delete b; return fail(WBC_E_ARG, "wbc_batch_create: model %d is null", i); }
    b->models[i] = models[i];
    dm[i] = models[i]->dev;
    if (models[i]->blob.njoints > b->max_nj) b->max_nj = models[i]->blob.njoints;
    if (models[i]->blob.nframes > b->max_nf) b->max_nf = models[i]->blob.nframes;
  }
  // everything allocated so far is released on any failure below (wbc_batch_destroy frees what is non-null)
#define HIP_TRY_B(expr)                                                                                              \
  do {                                                                                                               \
    hipError_t e_ = (expr);                                                                                          \
    if (e_ != hipSuccess) { wbc_batch_destroy(b); return fail(WBC_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); }  \
  } while (0)
  hipDeviceProp_t prop;
  HIP_TRY_B(hipGetDeviceProperties(&prop, device_id));
  const int per_cu = (int)(prop.maxSharedMemoryPerMultiProcessor / (size_t)tick_lds_bytes());
  b->grid = prop.multiProcessorCount * (per_cu < 1 ? 1 : (per_cu > 16 ? 16 : per_cu));
  if (n_models > 0) {   // n_models == 0: a QP-only handle (wbc_qp_solve / wbc_qp_solve_ls)
    HIP_TRY_B(hipMalloc((void**)&b->d_models, sizeof(DevModel) * n_models));
    HIP_TRY_B(hipMalloc((void**)&b->d_cfgs, sizeof(WbcConfig) * n_models));
    HIP_TRY_B(hipMalloc((void**)&b->d_plans, sizeof(DevPlan) * n_models));
    HIP_TRY_B(hipMemset(b->d_plans, 0, sizeof(DevPlan) * n_models));
    HIP_TRY_B(hipMemcpy(b->d_models, dm.data(), sizeof(DevModel) * n_models, hipMemcpyHostToDevice));
  }
#undef HIP_TRY_B
  *out = b;
  return WBC_OK;
}

extern "C" void wbc_batch_destroy(WbcBatch* b) {
  if (!b) return;
  (void)hipSetDevice(b->device_id);
  if (b->d_models) (void)hipFree(b->d_models);
  if (b->d_cfgs) (void)hipFree(b->d_cfgs);
  if (b->d_plans) (void)hipFree(b->d_plans);
  if (b->ws) (void)hipFree(b->ws);
  if (b->d_prof) (void)hipFree(b->d_prof);
  if (b->d_pu) (void)hipFree(b->d_pu);
  if (b->d_pq) (void)hipFree(b->d_pq);
  if (b->d_roll) (void)hipFree(b->d_roll);
  if (b->d_status) (void)hipFree(b->d_status);
  if (b->d_defer) (void)hipFree(b->d_defer);
  if (b->d_dstat) (void)hipFree(b->d_dstat);
  delete b;
}

static int rows_task(const WbcConfig& c, int* mcart) {
  int m = 0;
  for (int i = 0; i < WBC_NEE; ++i) m += c.task_ee[i] ? 6 : 0;
  m += c.task_trunk ? 6 : 0;
  m += c.task_com ? 3 : 0;
  *mcart = m;
  return m + (c.task_joint ? WBC_V_STRIDE : 0);
}
static int rows_con(const WbcConfig& c) {
  int p = (c.con_com ? 2 : 0) + (c.con_trunk ? 4 : 0);
  for (int i = 0; i < WBC_NEE; ++i) p += c.con_ee[i] ? 3 : 0;
  return p;
}

static int any_orth_plan(const WbcBatch* b) {
  for (int i = 0; i < b->n_models; ++i) if (b->plan_host[i].orth) return 1;
  return 0;
}

// Which stance feet's contact equalities the tick kernel eliminates structurally (contact_presolve in wbc_common.h).
// Enabled only when (1) every contact foot's rows are supported by the 6 base DoF + 3 own leg DoF, the leg sets disjoint,
// (2) NO active task touches an eliminated leg DoF (then H_ll = d^2 I, H_lf = 0 and the reduction costs no accuracy),
// (3) the reduced problem fits qp_core<16>.
// qpJointb "MANI"/"HYBRID" (Robot_Wrapper4.py:1220-1260) analysed on the tree: DoF i differentiates joint_id with respect to
// q[qi]; that finite difference is exactly zero unless q[qi] belongs to the free-flyer or to a PROPER ancestor of joint_id
// (wbc_posture_kernel makes the same test per sweep). If no sweep matters the tick kernels need no posture kernel at all.
// The sweeps of qpJointb "MANI" / "HYBRID" that matter, for wbc_posture_par_kernel (same loop, same skips, same `matters` test as
// wbc_posture_kernel): one record per sweep, with the configuration entries the reference's loop has left perturbed before it.
static void build_posture_par_plan(const DevModel& M, const WbcConfig& c, DevPlan* P) {
  P->mp_ok = 0; P->mp_n = 0; P->mp_all = 0; P->mp_prevmode = 0;
  const int mode = c.task_joint, literal = c.posture_literal;
  if (mode != WBC_JOINT_MANI && mode != WBC_JOINT_HYBRID) return;
  if (M.njoints > 22 || M.maxdepth > 8) return;
  uint32_t prev = 0, swept = 0;
  int n = 0;
  for (int i = 0; i < M.nv; ++i) {
    int joint_id;
    if (mode == WBC_JOINT_MANI) joint_id = (i < 6) ? 1 : (literal ? i + 1 - 5 : i - 4);
    else { joint_id = i - 6; if (joint_id < c.arm_base_id) continue; }
    if (joint_id >= M.njoints) continue;
    const int qi = literal ? i : ((i < 6) ? i : i + 1);
    swept |= 1u << i;                                            // (a sweep that does not matter still sets u_i = 0.5 (0 - 0) / d = 0)
    int jp = 1;
    for (int j = 2; j < M.njoints; ++j) if (M.idx_q[j] == qi) jp = j;
    const bool matters = (jp == 1) || (jp != joint_id && ((M.col_subtree[M.idx_v_of[jp]] >> joint_id) & 1u));
    if (matters) {
      if (n >= 32) return;
      P->mp_i[n] = i; P->mp_qi[n] = qi; P->mp_joint[n] = joint_id; P->mp_prev[n] = literal ? prev : 0u;
      int chain[8], len = 0;
      for (int j = joint_id; j > 1; j = M.parent[j]) { if (len >= 8) return; chain[len++] = j; }
      for (int k = 0; k < 8; ++k) P->mp_chain[n][k] = (k < len) ? chain[len - 1 - k] : -1;
      ++n;
    }
    if (literal) prev |= 1u << qi;
  }
  P->mp_n = n; P->mp_all = literal ? prev : 0u;
  if (mode == WBC_JOINT_HYBRID) P->mp_prevmode = ((M.nv >= 32) ? 0xFFFFFFFFu : ((1u << M.nv) - 1u)) & ~swept;   // DoF whose u stays the PREV value
  P->mp_ok = 1;
}

static void build_posture_plan(const DevModel& M, const WbcConfig& c, DevPlan* P) {
  const int mode = c.task_joint;
  if (mode != WBC_JOINT_MANI && mode != WBC_JOINT_HYBRID) return;
  build_posture_par_plan(M, c, P);
  const int literal = c.posture_literal;
  uint32_t zero = 0, pert = 0;
  for (int i = 0; i < M.nv; ++i) {
    int joint_id;
    if (mode == WBC_JOINT_MANI) joint_id = (i < 6) ? 1 : i - 4;
    else { joint_id = i - 6; if (joint_id < c.arm_base_id) continue; }
    if (joint_id >= M.njoints) continue;
    const int qi = literal ? i : ((i < 6) ? i : i + 1);
    if (qi < 7) return;                                    // the free-flyer moves everything: the sweeps are needed
    int jp = -1;
    for (int j = 2; j < M.njoints; ++j) if (M.idx_q[j] == qi) jp = j;
    if (jp < 0) return;
    if (jp != joint_id && ((M.col_subtree[M.idx_v_of[jp]] >> joint_id) & 1u)) return;   // a proper ancestor: matters
    zero |= 1u << i;
    if (literal) pert |= 1u << qi;
  }
  if (mode == WBC_JOINT_MANI) return;                      // (MANI starts from u = 0, not PREV; it never gets here anyway: i < 6)
  P->post_static = 1; P->post_zero = zero; P->post_pert = pert;
  // does an active constraint depend on a perturbed joint? (then the second FK pass is needed)
  uint32_t dofs = 0;
  for (int k = 0; k < M.nv; ++k) if ((pert >> M.col_q[k]) & 1u) dofs |= 1u << k;
  bool dep = c.con_com != 0 && dofs != 0;
  if (c.con_trunk && (M.frame_support[WBC_FR_TRUNK] & dofs)) dep = true;
  for (int e = 0; e < WBC_NEE; ++e) if (c.con_ee[e] && (M.frame_support[WBC_FR_EE0 + e] & dofs)) dep = true;
  P->post_fk2 = dep ? 1 : 0;
}

// The packed FK schedule shared by wbc_tick_sim3p_kernel and wbc_update_packed_kernel: the joints of tree depth 2 + L (L < 5) up to need_depth,
// one per lane-in-instance, and the q index of every revolute joint it reaches. Returns false if a level holds more than 16 joints.
static bool build_pk_fk(const DevModel& M, int need_depth, DevPlan* P) {
  for (int i = 0; i < 32; ++i) P->pk_scq[i] = -1;
  for (int L = 0; L < 5; ++L) {
    int cnt = 0;
    for (int i = 0; i < 16; ++i) { memset(&P->pk_fk[L][i], 0, sizeof P->pk_fk[L][i]); P->pk_fk[L][i].joint = -1; }
    for (int j = 2; j < M.njoints; ++j)
      if (M.depth[j] == L + 2 && M.depth[j] <= need_depth) {
        if (cnt < 16) {
          DevPlan::PkJoint& r = P->pk_fk[L][cnt];
          const bool rev = M.jtype[j] >= WBC_JT_RX && M.jtype[j] <= WBC_JT_RZ;
          r.joint = j; r.parent = M.parent[j]; r.a0 = 3 * M.ax0[j]; r.a1 = 3 * M.ax1[j]; r.a2 = 3 * M.ax2[j];
          r.rev = rev ? 1 : 0; r.q_idx = M.idx_q[j]; r.t0 = M.tp[j][0]; r.t1 = M.tp[j][1]; r.t2 = M.tp[j][2];
          if (rev && j < 32) P->pk_scq[j] = M.idx_q[j];
        }
        ++cnt;
      }
    if (cnt > 16) return false;
  }
  return true;
}

// Tables the packed orth and the packed box kernels share: the FK schedule over EVERY joint (tree depth 2 + L, L < 6, one joint per
// lane-in-instance), the revolute joints' q indices, body masses / centres, and per DoF its Jacobian column, the EE frames it moves and
// (need_subtree: the CoM Jacobian) the contiguous joint range of its subtree. Returns false if the model does not fit the layout.
static bool build_q_tables(const DevModel& M, DevPlan* P, bool need_subtree) {
  if (M.njoints > 22 || M.maxdepth > 7) return false;
  for (int i = 0; i < 32; ++i) { P->q_scq[i] = -1; memset(&P->q_dof[i], 0, sizeof P->q_dof[i]); P->q_dof[i].bl = P->q_dof[i].red = -1; memset(&P->q_jm[i], 0, sizeof P->q_jm[i]); }
  for (int L = 0; L < 6; ++L) {
    int cnt = 0;
    for (int i = 0; i < 16; ++i) { memset(&P->q_fk[L][i], 0, sizeof P->q_fk[L][i]); P->q_fk[L][i].joint = -1; }
    for (int j = 2; j < M.njoints; ++j)
      if (M.depth[j] == L + 2) {
        if (cnt >= 16) return false;
        DevPlan::PkJoint& r = P->q_fk[L][cnt++];
        const bool rev = M.jtype[j] >= WBC_JT_RX && M.jtype[j] <= WBC_JT_RZ;
        r.joint = j; r.parent = M.parent[j]; r.a0 = 3 * M.ax0[j]; r.a1 = 3 * M.ax1[j]; r.a2 = 3 * M.ax2[j];
        r.rev = rev ? 1 : 0; r.q_idx = M.idx_q[j]; r.t0 = M.tp[j][0]; r.t1 = M.tp[j][1]; r.t2 = M.tp[j][2];
        if (rev) P->q_scq[j] = M.idx_q[j];
      }
  }
  for (int j = 1; j < M.njoints; ++j) { P->q_jm[j].m = M.mass[j]; P->q_jm[j].c0 = M.com[j][0]; P->q_jm[j].c1 = M.com[j][1]; P->q_jm[j].c2 = M.com[j][2]; }
  for (int d = 0; d < M.nv; ++d) {
    DevPlan::QDof& r = P->q_dof[d];
    const int j = M.col_joint[d];
    r.joint = j; r.lin = M.col_lin[d]; r.ang = M.col_ang[d];
    const uint32_t sub = M.col_subtree[d];
    int lo = -1, hi = -1, cntj = 0;
    for (int k = 1; k < M.njoints; ++k) if ((sub >> k) & 1u) { if (lo < 0) lo = k; hi = k; ++cntj; }
    if (need_subtree) {
      if (lo < 0 || hi - lo + 1 != cntj) return false;                    // not contiguous
      if (j != 1 && cntj > 8) return false;                               // the kernel's sub-tree loop
    }
    r.sub_lo = lo; r.sub_hi = hi;
    for (int e = 0; e < WBC_NEE; ++e) if ((M.frame_support[WBC_FR_EE0 + e] >> d) & 1u) r.supmask |= 1u << e;
  }
  return true;
}

// The packed orth kernel's plan (wbc_tick_orthp_kernel): equality-only task problems — the only constraints are the eliminated stance
// feet's contact rows (no trunk / CoM box, no velocity box), tasks = any EE tasks + optionally the CoM task + posture Tikhonov / PREV.
static void build_orthp_plan(const DevModel& M, const WbcConfig& c, DevPlan* P) {
  P->q_ok = 0;
  const int nelim = P->nelim, nl = 3 * nelim;
  if (!P->orth || nelim < 1 || nelim > 4) return;
  // INEQ: the kernel's variant with inequality rows (trunk box, CoM box, the velocity box of every DoF — in the reduced coordinates rows of Z) and a
  // trunk task: everything beyond the equality-only family of BASELINE configs[1]
  const bool ineq = c.use_bounds || c.con_trunk || c.con_com || c.task_trunk;
  const bool trunk_is_root = M.frame_joint[WBC_FR_TRUNK] == 1 && M.frame_p[WBC_FR_TRUNK][0] == 0 && M.frame_p[WBC_FR_TRUNK][1] == 0 && M.frame_p[WBC_FR_TRUNK][2] == 0;
  if (ineq && (!c.use_bounds || c.con_ee[4] || ((c.con_trunk || c.task_trunk) && !trunk_is_root) || P->n_red > 12)) return;
  if (P->p_keep != (ineq ? (c.con_com ? 2 : 0) + (c.con_trunk ? 4 : 0) : 0)) return;
  if (c.task_joint != WBC_JOINT_TIKHONOV && c.task_joint != WBC_JOINT_PREV) return;
  if (P->n_red > 15) return;
  uint32_t lockmask = 0;
  if (c.use_bounds) for (int d = c.lock_from; d < M.nv; ++d) if (d >= 6 && P->lidx[d] < 0) lockmask |= 1u << d;
  if (!build_q_tables(M, P, true)) return;
  for (int i = 0; i < 18; ++i) P->q_bl2dof[i] = 0;
  for (int i = 0; i < 16; ++i) P->q_red2dof[i] = 0;
  for (int e = 0; e < 8; ++e) P->q_efoot[e] = -1;
  // DoF records: [base; eliminated legs] positions, reduced (free) variables 6.., subtree = a contiguous joint range (depth-first numbering)
  int nred = 6;
  uint32_t freemask = 0;
  for (int d = 0; d < M.nv; ++d) {
    DevPlan::QDof& r = P->q_dof[d];
    if (d < 6) r.bl = d;
    else if (P->lidx[d] >= 0) r.bl = 6 + P->lidx[d];
    else if ((lockmask >> d) & 1u) { }                              // locked at 0 by the velocity box: its column leaves the problem
    else { if (nred >= 15) return; r.red = nred; P->q_red2dof[nred++] = d; freemask |= 1u << d; }
    if (r.bl >= 0) P->q_bl2dof[r.bl] = d;
    DevPlan::XVar& v = P->q_dmp[d];
    memset(&v, 0, sizeof v);
    v.dof = d; v.dq_idx = c.damper_qidx[d]; v.d_lo = c.damper_lo[d]; v.d_hi = c.damper_hi[d]; v.d_vm = c.damper_vmax[d];
  }
  if (nred != P->n_red) return;
  P->q_nred = nred;
  // EE tasks: support = base + (one eliminated foot's own leg | free variables)
  P->q_armsup = 0;
  uint32_t legall = 0;
  for (int l = 0; l < nl; ++l) legall |= 1u << P->legd[l];
  for (int e = 0; e < WBC_NEE; ++e) {
    if (!c.task_ee[e]) continue;
    const uint32_t sup = M.frame_support[WBC_FR_EE0 + e];
    if (M.depth[M.frame_joint[WBC_FR_EE0 + e]] > 7) return;
    if (sup & freemask) P->q_armsup |= 1u << e;
    const uint32_t legs = sup & legall;
    if (sup & ~(0x3Fu | legall | freemask | lockmask)) return;
    if (legs) {
      int f = -1;
      for (int t = 0; t < nelim; ++t) {
        const uint32_t own = (1u << P->legd[3 * t]) | (1u << P->legd[3 * t + 1]) | (1u << P->legd[3 * t + 2]);
        if (legs == own) f = t;
      }
      if (f < 0) return;
      P->q_efoot[e] = f;
    }
  }
  P->q_ok = ineq ? 2 : 1;
  P->q_nlock = __builtin_popcount(lockmask);
  // roll-outs of these configurations update their state on wbc_update_packed_kernel too: its FK schedule down to the deepest frame the estimator reads
  int need = M.depth[M.frame_joint[WBC_FR_TRUNK]];
  for (int e = 0; e < 5; ++e) if (M.depth[M.frame_joint[WBC_FR_EE0 + e]] > need) need = M.depth[M.frame_joint[WBC_FR_EE0 + e]];
  P->pk_update_ok = (need <= 6 && build_pk_fk(M, need, P)) ? 1 : 0;
}

// The packed box kernel's plan (wbc_tick_boxp_kernel): task problems without a single constraint row (the warm-up problem of setInitialState,
// Robot_Wrapper4.py:196-351): trunk task (trunk frame = the free-flyer's placement) and / or EE tasks + posture Tikhonov / PREV, velocity box on.
// Eliminated: the six base DoF and, if more than 16 bounded DoF remain, those with the widest box (position range x velocity limit of the damper
// entry the DoF looks at): their bounds are checked after the solve, a violation sends the instance to the general path. Every limb DoF must
// move at most ONE active task's frame (the block-arrow structure of H).
static void build_boxp_plan(const DevModel& M, const WbcConfig& c, int prows, DevPlan* P) {
  P->x_ok = 0;
  if (prows != 0 || !c.use_bounds || c.task_com) return;
  if (c.task_joint != WBC_JOINT_TIKHONOV && c.task_joint != WBC_JOINT_PREV) return;
  const bool trunk_is_root = M.frame_joint[WBC_FR_TRUNK] == 1 && M.frame_p[WBC_FR_TRUNK][0] == 0 && M.frame_p[WBC_FR_TRUNK][1] == 0 && M.frame_p[WBC_FR_TRUNK][2] == 0;
  if (c.task_trunk && !trunk_is_root) return;
  if (M.nv > 26 || !build_q_tables(M, P, false)) return;
  for (int d = 0; d < M.nv; ++d) if (M.col_joint[d] == 1 ? d >= 6 : (M.col_q[d] < 0 || M.col_q[d] >= M.nq)) return;   // one free-flyer + 1-DoF joints
  for (int e = 0; e < WBC_NEE; ++e) if (c.task_ee[e] && M.depth[M.frame_joint[WBC_FR_EE0 + e]] > 7) return;
  uint32_t lockmask = 0, tmask = 0;
  for (int d = c.lock_from; d < M.nv; ++d) if (d >= 6) lockmask |= 1u << d;
  for (int e = 0; e < WBC_NEE; ++e) if (c.task_ee[e]) tmask |= 1u << e;
  int freed[32], nfree = 0;
  for (int d = 6; d < M.nv; ++d) {
    if ((lockmask >> d) & 1u) continue;
    if (__builtin_popcount(P->q_dof[d].supmask & tmask) > 1) return;
    freed[nfree++] = d;
  }
  int extra = nfree - 16;
  if (extra > 2) return;
  for (int i = 0; i < 32; ++i) P->x_role[i] = -1;
  for (int i = 0; i < 16; ++i) { memset(&P->x_kept[i], 0, sizeof P->x_kept[i]); P->x_kept[i].task = -1; P->x_limb[i] = 1u << i; }
  for (int i = 0; i < 8; ++i) { memset(&P->x_elim[i], 0, sizeof P->x_elim[i]); P->x_elim[i].task = -1; }
  auto fill = [&](DevPlan::XVar& v, int d) {
    v.dof = d; v.dq_idx = c.damper_qidx[d]; v.d_lo = c.damper_lo[d]; v.d_hi = c.damper_hi[d]; v.d_vm = c.damper_vmax[d];
    const uint32_t tk = P->q_dof[d].supmask & tmask;
    v.task = (d >= 6 && tk) ? __builtin_ctz(tk) : -1;
  };
  int ne = 6;
  for (int d = 0; d < 6; ++d) { P->x_role[d] = d; fill(P->x_elim[d], d); }
  uint32_t elim_extra = 0;
  for (; extra > 0; --extra) {
    int best = -1; double bs = -1.0;
    for (int i = 0; i < nfree; ++i) {
      const int d = freed[i];
      if ((elim_extra >> d) & 1u) continue;
      const double sc = (c.damper_hi[d] - c.damper_lo[d]) * c.damper_vmax[d];
      if (sc > bs) { bs = sc; best = d; }
    }
    if (best < 0) return;
    elim_extra |= 1u << best;
    P->x_role[best] = ne; fill(P->x_elim[ne], best); ++ne;
  }
  int nk = 0;
  for (int i = 0; i < nfree; ++i) {
    const int d = freed[i];
    if ((elim_extra >> d) & 1u) continue;
    P->x_role[d] = 16 + nk; fill(P->x_kept[nk], d); ++nk;
  }
  for (int k = 0; k < nk; ++k)
    for (int k2 = 0; k2 < nk; ++k2)
      if (P->x_kept[k].task >= 0 && P->x_kept[k].task == P->x_kept[k2].task) P->x_limb[k] |= 1u << k2;
  P->x_ne = ne; P->x_nk = nk; P->x_nlock = __builtin_popcount(lockmask);
  P->x_ok = 1;
  // the warm-up roll-out (mode WBC_ROLLOUT_WARMUP) updates its state on wbc_update_packed_kernel too: its FK schedule down to the deepest frame it reads
  int need = M.depth[M.frame_joint[WBC_FR_TRUNK]];
  for (int e = 0; e < 5; ++e) if (M.depth[M.frame_joint[WBC_FR_EE0 + e]] > need) need = M.depth[M.frame_joint[WBC_FR_EE0 + e]];
  P->pk_update_ok = (need <= 6 && build_pk_fk(M, need, P)) ? 1 : 0;
}

static void build_plan(const DevModel& M, const WbcConfig& c, int prows, DevPlan* P) {
  memset(P, 0, sizeof *P);
  build_posture_plan(M, c, P);
  for (int e = 0; e < WBC_NEE; ++e) { if (c.task_ee[e]) P->task_ee_mask |= 1u << e; if (c.con_ee[e]) P->con_ee_mask |= 1u << e; }
  P->flags = (c.con_com ? 1u : 0u) | (c.con_trunk ? 2u : 0u) | (c.task_trunk ? 4u : 0u) | (c.use_bounds ? 8u : 0u) | ((uint32_t)(c.task_joint & 7) << 4);
  for (int i = 0; i < 32; ++i) { P->pos[i] = -1; P->lidx[i] = -1; }
  uint32_t legmask = 0;
  int prow = (c.con_com ? 2 : 0) + (c.con_trunk ? 4 : 0), nelim = 0, l = 0;
  for (int e = 0; e < 4; ++e) {
    if (!c.con_ee[e]) continue;
    const uint32_t sup = M.frame_support[WBC_FR_EE0 + e], legs = sup & ~0x3Fu;
    if ((sup & 0x3Fu) != 0x3Fu || __builtin_popcount(legs) != 3 || (legs & legmask)) return;
    legmask |= legs;
    P->elimrows |= 7u << prow;
    P->rowstart[nelim++] = prow;
    for (int d = 0; d < M.nv; ++d) if ((legs >> d) & 1u) { P->lidx[d] = l; P->legd[l++] = d; }
    prow += 3;
  }
  if (!nelim && c.task_joint) build_boxp_plan(M, c, prows, P);
  if (!nelim || !c.task_joint) return;
  // `clean`: no task touches the stance legs (H_ll = d^2 I, H_lf = 0): the explicit G = -K^-1 B costs no accuracy (contact_presolve,
  // the sim3 kernels). Otherwise the elimination goes through an orthonormal null-space basis (contact_presolve_orth).
  bool clean = !c.task_com;
  for (int e = 0; e < WBC_NEE; ++e) if (c.task_ee[e] && (M.frame_support[WBC_FR_EE0 + e] & legmask)) clean = false;
  if (c.task_trunk && (M.frame_support[WBC_FR_TRUNK] & legmask)) clean = false;
  // DoF the velocity box locks at 0 (lb = ub = 0 from lock_from on, Robot_Wrapper4.py:627-630) are known: they leave the reduced
  // problem altogether (their q̇ is 0, they contribute to nothing else) — which also leaves room in qp_core<16> for the
  // extra unknown a rank-deficient stance-leg block keeps (pivoted elimination in wbc_tick_sim3_kernel)
  uint32_t lockmask = 0;
  if (c.use_bounds) for (int d = c.lock_from; d < M.nv; ++d) if (d >= 6 && !((legmask >> d) & 1u)) lockmask |= 1u << d;
  // (exact whatever rows touch them: a column that multiplies a velocity fixed at 0 contributes to nothing)
  const int nlock = __builtin_popcount(lockmask);
  const int n_red = M.nv - 3 * nelim - nlock, p_keep = prows - 3 * nelim;
  if (n_red > WBC_PLAN_NR || n_red < 6 || p_keep + (c.use_bounds ? 3 * nelim + (clean ? 0 : 6) : 0) > WBC_MAX_P) return;
  int cnt = 0;
  for (int d = 0; d < M.nv; ++d) if (!(((legmask | lockmask) >> d) & 1u)) { P->pos[d] = cnt; P->Fd[cnt++] = d; }
  P->nlock = nlock;
  for (int f = 0; f < M.nframes; ++f)
    for (int d = 0; d < M.nv; ++d)
      if (((M.frame_support[f] >> d) & 1u) && P->pos[d] >= 0) P->redsup[f] |= 1u << P->pos[d];
  // kept rows in findConstraints order: CoM (whole-body support), trunk box (trunk frame), Grip contact
  {
    int r = 0;
    if (c.con_com) { P->legrows |= 3u << r; r += 2; }
    if (c.con_trunk) { if (M.frame_support[WBC_FR_TRUNK] & legmask) P->legrows |= 15u << r; r += 4; }
    for (int e = 0; e < WBC_NEE; ++e) {
      if (!c.con_ee[e]) continue;
      if (!((P->elimrows >> r) & 1u) && (M.frame_support[WBC_FR_EE0 + e] & legmask)) P->legrows |= 7u << r;
      r += 3;
    }
  }
  P->nelim = nelim; P->n_red = n_red; P->p_keep = p_keep;
  if (!clean) { P->orth = 1; build_orthp_plan(M, c, P); return; }
  P->enabled = 1;
  // ---- the packed kernel (wbc_tick_sim3p_kernel) covers the sim3 switch-set family only: Grip task or none, no trunk / CoM
  // task, the kept rows = the trunk box (base support only), velocity bounds on, a posture mode it can form itself, every
  // joint it needs within tree depth 6 and at most 16 joints per level
  // (a trunk task — base support only — runs on the kernel's TRUNK variant)
  const bool trunk_is_root = M.frame_joint[WBC_FR_TRUNK] == 1 && M.frame_p[WBC_FR_TRUNK][0] == 0 && M.frame_p[WBC_FR_TRUNK][1] == 0 && M.frame_p[WBC_FR_TRUNK][2] == 0;
  bool ok = c.use_bounds && !(c.task_trunk && !trunk_is_root) && !c.task_com && !c.con_com && !c.con_ee[4] && p_keep == (c.con_trunk ? 4 : 0) &&
            n_red <= 12 && (P->task_ee_mask & ~16u) == 0 && P->legrows == 0 &&
            true;
  // the posture modes the packed kernel forms itself; any other (MANI / HYBRID with sweeps that matter, CUSTOM) runs on its QCON variant with
  // the posture kernel's (or the caller's) posture_u / q_con — without the trunk task
  const bool own_mode = c.task_joint == WBC_JOINT_TIKHONOV || c.task_joint == WBC_JOINT_PREV ||
                        (c.task_joint >= WBC_JOINT_MANI && c.task_joint <= WBC_JOINT_HYBRID && P->post_static && !P->post_fk2);
  int need_depth = 0;
  for (int k = 0; k < n_red; ++k) if (M.depth[M.col_joint[P->Fd[k]]] > need_depth) need_depth = M.depth[M.col_joint[P->Fd[k]]];
  for (int l = 0; l < 3 * nelim; ++l) if (M.depth[M.col_joint[P->legd[l]]] > need_depth) need_depth = M.depth[M.col_joint[P->legd[l]]];
  if (M.depth[M.frame_joint[WBC_FR_EE0 + 4]] > need_depth) need_depth = M.depth[M.frame_joint[WBC_FR_EE0 + 4]];
  if (need_depth > 6) ok = false;
  for (int i = 0; i < 32; ++i) P->pk_scq[i] = -1;
  if (ok && !build_pk_fk(M, need_depth, P)) ok = false;
  for (int i = 0; i < 16; ++i) {
    memset(&P->pk_var[i], 0, sizeof P->pk_var[i]);
    memset(&P->pk_leg[i], 0, sizeof P->pk_leg[i]);
    for (int pass = 0; pass < 2; ++pass) {
      DevPlan::PkCol& r = pass ? P->pk_leg[i] : P->pk_var[i];
      const bool on = pass ? (i < 3 * nelim) : (i < n_red);
      const int d = on ? (pass ? P->legd[i] : P->Fd[i]) : 0;
      r.dof = d; r.joint = M.col_joint[d]; r.lin = M.col_lin[d]; r.ang = M.col_ang[d];
      r.dq_idx = c.damper_qidx[d]; r.d_lo = c.damper_lo[d]; r.d_hi = c.damper_hi[d]; r.d_vm = c.damper_vmax[d];
    }
  }
  P->packed_ok = (ok && own_mode) ? 1 : 0;
  P->packed_ok_pu = (ok && !c.task_trunk) ? 1 : 0;
  bool upd = ok;   // (any posture mode: the state update does not depend on it; the trunk reference state is advanced too since round 3)
  for (int e = 0; e < 5; ++e) if (M.depth[M.frame_joint[WBC_FR_EE0 + e]] > need_depth) upd = false;
  if (M.depth[M.frame_joint[WBC_FR_TRUNK]] > need_depth) upd = false;
  P->pk_update_ok = upd ? 1 : 0;
}

extern "C" int wbc_batch_configure(WbcBatch* b, int mi, const WbcConfig* cfg) {
  if (!b || !cfg || mi < 0 || mi >= b->n_models) return fail(WBC_E_ARG, "wbc_batch_configure: bad arguments");
  const int nv = b->models[mi]->blob.nv, nq = b->models[mi]->blob.nq;
  if (cfg->task_joint < 0 || cfg->task_joint > WBC_JOINT_CUSTOM) return fail(WBC_E_ARG, "unknown posture mode %d", cfg->task_joint);
  if (cfg->task_joint == WBC_JOINT_HYBRID && (cfg->arm_base_id < 1 || cfg->arm_base_id >= b->models[mi]->blob.njoints))
    return fail(WBC_E_ARG, "HYBRID posture: arm_base_id %d is not a joint of the model", cfg->arm_base_id);
  if (!cfg->task_joint) return fail(WBC_E_UNSUPPORTED, "the posture task must be on: without it H = J'J is singular (Robot_Wrapper4.py:1199-1206)");
  for (int i = 0; i < nv; ++i)
    if (cfg->use_bounds && (cfg->damper_qidx[i] < 0 || cfg->damper_qidx[i] >= nq)) return fail(WBC_E_ARG, "damper_qidx[%d] = %d out of range", i, cfg->damper_qidx[i]);
  int mcart = 0;
  const int m = rows_task(*cfg, &mcart), p = rows_con(*cfg);
  if (p > WBC_MAX_P) return fail(WBC_E_ARG, "too many constraint rows (%d)", p);
  // (all models of a batch must share the task / constraint switches; that is checked when the batch is USED, so that a
  //  configured multi-model batch can be moved to another switch set one model at a time — same_switches below)
  HIP_TRY(hipSetDevice(b->device_id));
  b->cfg_host[mi] = *cfg;
  b->configured[mi] = true;
  b->plan_host[mi] = DevPlan();
  b->mrows = m; b->prows = p; b->mcart = mcart;
  HIP_TRY(hipMemcpy(b->d_cfgs + mi, cfg, sizeof *cfg, hipMemcpyHostToDevice));
  DevPlan plan;
  build_plan(b->models[mi]->dev, *cfg, p, &plan);
  b->plan_host[mi] = plan;
  HIP_TRY(hipMemcpy(b->d_plans + mi, &plan, sizeof plan, hipMemcpyHostToDevice));
  return WBC_OK;
}

extern "C" int wbc_task_rows(const WbcBatch* b) { return b ? b->mrows : WBC_E_ARG; }
extern "C" int wbc_constraint_rows(const WbcBatch* b) { return b ? b->prows : WBC_E_ARG; }

extern "C" int wbc_batch_set_option(WbcBatch* b, const char* name, int value) {
  if (!b || !name) return fail(WBC_E_ARG, "wbc_batch_set_option: null");
  if (!strcmp(name, "jtj_mfma")) { b->jtj_mfma = value < 0 ? -1 : (value ? 1 : 0); return WBC_OK; }
  if (!strcmp(name, "presolve")) { b->presolve = value; return WBC_OK; }
  if (!strcmp(name, "presolve_orth")) { b->presolve_orth = value; return WBC_OK; }
  if (!strcmp(name, "orth_qr")) { b->orth_qr = value; return WBC_OK; }
  if (!strcmp(name, "packed_update")) { b->packed_update = value; return WBC_OK; }
  if (!strcmp(name, "presolve_tol_exp")) { double t = 1.0; for (int i = 0; i < value; ++i) t *= 0.1; b->sing_tol = t; return WBC_OK; }
  if (!strcmp(name, "sim3_kernel")) { b->sim3_kernel = value; return WBC_OK; }
  if (!strcmp(name, "packed_kernel")) { b->packed_kernel = value; return WBC_OK; }
  if (!strcmp(name, "packed_min_batch")) { b->packed_min_batch = value < 1 ? 1 : value; return WBC_OK; }
  if (!strcmp(name, "packed_orth")) { b->packed_orth = value; return WBC_OK; }
  if (!strcmp(name, "packed_box")) { b->packed_box = value; return WBC_OK; }
  if (!strcmp(name, "posture_par")) { b->posture_par = value; return WBC_OK; }
  if (!strcmp(name, "dbg_alias_inputs")) { b->dbg_alias = value; return WBC_OK; }
  if (!strcmp(name, "warm_start")) { b->warm_start = value != 0; return WBC_OK; }
  if (!strcmp(name, "refine")) { b->refine = value > 0; return WBC_OK; }
  if (!strcmp(name, "count_pivoted")) { b->count_pivoted = value != 0; return WBC_OK; }
  if (!strcmp(name, "dbg_force_defer")) { b->force_defer = value != 0; return WBC_OK; }
  if (!strcmp(name, "dbg_stop")) {
#ifdef WBC_ABLATE
    b->dbg_stop = value; return WBC_OK;
#else
    return value == 0 ? WBC_OK : fail(WBC_E_UNSUPPORTED, "dbg_stop needs the ablation build of the library (make -C csrc ablate; WBC_HIP_LIB=.../libwbc_hip_ablate.so)");
#endif
  }
  if (!strcmp(name, "grid")) { if (value < 1) return fail(WBC_E_ARG, "grid must be >= 1"); b->grid = value; return WBC_OK; }
  return fail(WBC_E_ARG, "unknown option %s", name);
}

extern "C" int wbc_batch_get_stat(WbcBatch* b, const char* name, void* stream, int64_t* out) {
  if (!b || !name || !out) return fail(WBC_E_ARG, "wbc_batch_get_stat: null");
  HIP_TRY(hipSetDevice(b->device_id));
  if (!strcmp(name, "last_path")) { *out = b->last_path; return WBC_OK; }
  if (!strcmp(name, "last_qp_path")) { *out = b->last_qp_path; return WBC_OK; }
  if (!strcmp(name, "last_update_packed")) { *out = b->last_update_packed; return WBC_OK; }
  if (!strcmp(name, "last_posture_par")) { *out = b->last_posture_par; return WBC_OK; }
  if (!strcmp(name, "last_orth")) { *out = (b->last_path == 0 || b->last_path == 3) ? b->last_orth : 0; return WBC_OK; }
  if (!strcmp(name, "deferred_last")) {      // waits for `stream`
    *out = 0;
    if (b->last_path >= 2) {                 // packed kernels: instances the tail redid on the general path in the last launch
      if (!b->d_dstat) return WBC_OK;
      unsigned long long v = 0;
      HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
      HIP_TRY(hipMemcpy(&v, b->d_dstat, sizeof v, hipMemcpyDeviceToHost));
      *out = ((uint32_t)(v >> 32) == b->tick_seq) ? (int64_t)(v & 0xFFFFFFFFull) : 0;
      return WBC_OK;
    }
    if (!b->d_defer || !b->last_path) return WBC_OK;
    int32_t c = 0;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemcpy(&c, b->d_defer + 1 + b->max_batch + 2, sizeof c, hipMemcpyDeviceToHost));
    *out = c;
    return WBC_OK;
  }
  if (!strcmp(name, "pivoted_last")) {       // needs option "count_pivoted"; waits for `stream`
    *out = 0;
    if (!b->d_defer || !b->last_path || !b->count_pivoted) return WBC_OK;
    int32_t c = 0;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemcpy(&c, b->d_defer + 1 + b->max_batch, sizeof c, hipMemcpyDeviceToHost));
    *out = c;
    return WBC_OK;
  }
  if (!strcmp(name, "sim3_lds_bytes")) { *out = sim3_lds_bytes(); return WBC_OK; }
  if (!strcmp(name, "orthp_lds_bytes")) { *out = orthp_lds_bytes(); return WBC_OK; }
  if (!strcmp(name, "tick_lds_bytes")) { *out = tick_lds_bytes(); return WBC_OK; }
  return fail(WBC_E_ARG, "unknown statistic %s", name);
}

extern "C" int wbc_batch_synchronize(WbcBatch* b, void* stream) {
  if (!b) return fail(WBC_E_ARG, "null batch");
  HIP_TRY(hipSetDevice(b->device_id));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return WBC_OK;
}

// ---------------------------------------------------------------------------------------------- staging
// mem == WBC_MEM_HOST: inputs are copied into the batch workspace, outputs copied back after the launch.
// mem == WBC_MEM_DEVICE: pointers pass through untouched (no copies, no synchronisation).
struct Stager {
  WbcBatch* b; int mem; hipStream_t s;
  struct Item { void** slot; const void* host; size_t bytes; bool out; };
  std::vector<Item> items;
  template <class T> void in(const T** slot, size_t count) { add((void**)slot, count * sizeof(T), false); }
  template <class T> void out(T** slot, size_t count) { add((void**)slot, count * sizeof(T), true); }
  void add(void** slot, size_t bytes, bool is_out) {
    if (mem == WBC_MEM_DEVICE || !*slot || !bytes) return;
    items.push_back({slot, *slot, bytes, is_out});
  }
  int stage() {
    if (mem == WBC_MEM_DEVICE) return WBC_OK;
    size_t total = 0;
    for (auto& it : items) total += (it.bytes + 255) & ~(size_t)255;
    if (total > b->ws_bytes) {
      if (b->ws) HIP_TRY(hipFree(b->ws));
      b->ws = nullptr; b->ws_bytes = 0;
      HIP_TRY(hipMalloc(&b->ws, total));
      b->ws_bytes = total;
    }
    size_t off = 0;
    for (auto& it : items) {
      void* d = (char*)b->ws + off;
      off += (it.bytes + 255) & ~(size_t)255;
      if (!it.out) HIP_TRY(hipMemcpyAsync(d, it.host, it.bytes, hipMemcpyHostToDevice, s));
      *it.slot = d;
    }
    return WBC_OK;
  }
  int finish() {
    if (mem == WBC_MEM_DEVICE) return WBC_OK;
    for (auto& it : items)
      if (it.out) HIP_TRY(hipMemcpyAsync((void*)it.host, *it.slot, it.bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return WBC_OK;
  }
};

// the switches that fix the row layout of A, C and which inputs a tick reads: identical for every model of a batch
static bool same_switches(const WbcConfig& a, const WbcConfig& c) {
  for (int i = 0; i < WBC_NEE; ++i)
    if ((a.task_ee[i] != 0) != (c.task_ee[i] != 0) || (a.con_ee[i] != 0) != (c.con_ee[i] != 0)) return false;
  return (a.task_trunk != 0) == (c.task_trunk != 0) && (a.task_com != 0) == (c.task_com != 0) &&
         (a.con_com != 0) == (c.con_com != 0) && (a.con_trunk != 0) == (c.con_trunk != 0) &&
         (a.use_bounds != 0) == (c.use_bounds != 0) && (a.task_joint != 0) == (c.task_joint != 0);
}
static int check_batch(WbcBatch* b, int B, const char* who, bool need_cfg, bool need_model = true) {
  if (!b) return fail(WBC_E_ARG, "%s: null batch", who);
  if (need_model && b->n_models < 1) return fail(WBC_E_STATE, "%s: this handle was created without a model", who);
  if (B < 1 || B > b->max_batch) return fail(WBC_E_ARG, "%s: B = %d outside [1, max_batch = %d]", who, B, b->max_batch);
  if (need_cfg) {
    for (int i = 0; i < b->n_models; ++i)
      if (!b->configured[i]) return fail(WBC_E_STATE, "%s: model %d has no configuration (wbc_batch_configure)", who, i);
    for (int i = 1; i < b->n_models; ++i)
      if (!same_switches(b->cfg_host[0], b->cfg_host[i]))
        return fail(WBC_E_STATE, "%s: all models of a batch must share the task/constraint switches (model %d differs from model 0)", who, i);
  }
  return WBC_OK;
}
static int grid_for(const WbcBatch* b, int B) { return B < b->grid ? B : b->grid; }   // persistent kernels (QP, integrate)
static int grid_tick(const WbcBatch*, int B) { return B; }                           // tick kernels: one instance per workgroup

static void stage_tick_in(Stager& st, WbcTickIn& in, int B, const WbcBatch* b) {
  const size_t n = (size_t)B;
  st.in(&in.q, n * WBC_Q_STRIDE);
  st.in(&in.ee_target, n * 15); st.in(&in.prev_ee_target, n * 15);
  st.in(&in.trunk_target, n * 3); st.in(&in.prev_trunk_target, n * 3);
  st.in(&in.trunk_box_center, n * 4);
  st.in(&in.ee_ref_rot, n * 45); st.in(&in.ee_prev_rot, n * 45);
  st.in(&in.trunk_ref_euler, n * 3); st.in(&in.trunk_prev_rot, n * 9);
  st.in(&in.com_target, n * 3); st.in(&in.com_target_vel, n * 3);
  st.in(&in.model_id, n);
  st.in(&in.posture_u, n * WBC_V_STRIDE); st.in(&in.q_con, n * WBC_Q_STRIDE);
  st.in(&in.working_set, n * 2);
  (void)b;
}

static int validate_tick_in(const WbcBatch* b, const WbcTickIn* in, const char* who) {
  const WbcConfig& c = b->cfg_host[0];
  if (!in || !in->q) return fail(WBC_E_ARG, "%s: q is required", who);
  bool any_ee = false;
  for (int i = 0; i < WBC_NEE; ++i) any_ee |= c.task_ee[i] != 0;
  if (any_ee && (!in->ee_target || !in->prev_ee_target)) return fail(WBC_E_ARG, "%s: EE tasks need ee_target and prev_ee_target", who);
  if ((in->ee_ref_rot != nullptr) != (in->ee_prev_rot != nullptr)) return fail(WBC_E_ARG, "%s: ee_ref_rot and ee_prev_rot go together", who);
  if (c.task_trunk && (!in->trunk_target || !in->prev_trunk_target || !in->trunk_ref_euler || !in->trunk_prev_rot))
    return fail(WBC_E_ARG, "%s: the trunk task needs trunk_target, prev_trunk_target, trunk_ref_euler, trunk_prev_rot", who);
  if (c.con_trunk && !in->trunk_box_center) return fail(WBC_E_ARG, "%s: the trunk constraint needs trunk_box_center", who);
  if (c.task_com && (!in->com_target || !in->com_target_vel)) return fail(WBC_E_ARG, "%s: the CoM task needs com_target and com_target_vel", who);
  if (b->n_models > 1 && !in->model_id) return fail(WBC_E_ARG, "%s: model_id is required with %d models", who, b->n_models);
  for (int i = 0; i < b->n_models; ++i)
    if (b->cfg_host[i].task_joint == WBC_JOINT_CUSTOM && !in->posture_u) return fail(WBC_E_ARG, "%s: posture mode CUSTOM needs posture_u", who);
  return WBC_OK;
}

// qpJointb "MANI" / "HYBRID" (Robot_Wrapper4.py:1220-1260) ahead of the tick kernel: fills a.in.posture_u (and a.in.q_con
// in literal mode) from the device-resident q unless the caller supplied them. Pointers in `a` are device pointers here.
static int run_posture(WbcBatch* b, int B, const double* q, const int32_t* model_id, double* u, double* q_after, void* stream) {
  PostureArgs pa;
  memset(&pa, 0, sizeof pa);
  pa.models = b->d_models; pa.cfgs = b->d_cfgs; pa.plans = b->d_plans; pa.B = B; pa.n_models = b->n_models; pa.q = q; pa.model_id = model_id; pa.u = u; pa.q_after = q_after;
  bool par = b->posture_par != 0;
  for (int i = 0; i < b->n_models && par; ++i) par = b->configured[i] && b->plan_host[i].mp_ok != 0;
  bool three = par && b->posture_par != 3;                 // three instances per wavefront where every model has at most 21 sweeps (option value 3: one per wavefront)
  for (int i = 0; i < b->n_models && three; ++i) three = b->plan_host[i].mp_n <= 21;
  b->last_posture_par = par ? (three ? 2 : 1) : 0;
  if (int e = par ? launch_posture_par(pa, B, stream, three) : launch_posture(pa, B, stream)) return fail(WBC_E_HIP, "posture kernel launch failed: %s", hipGetErrorString((hipError_t)e));
  return WBC_OK;
}
static int auto_posture(WbcBatch* b, KernelArgs& a, int B, void* stream) {
  bool need = false, literal = false;
  for (int i = 0; i < b->n_models; ++i) {
    const int tj = b->cfg_host[i].task_joint;
    if (tj == WBC_JOINT_MANI || tj == WBC_JOINT_HYBRID) { need = true; literal |= b->cfg_host[i].posture_literal != 0; }
  }
  if (!need || a.in.posture_u) return WBC_OK;
  if (b->presolve) {   // every sweep structurally zero on every model: the tick derives u and the leaked state itself
    bool all_static = true;
    for (int i = 0; i < b->n_models; ++i) {
      const int tj = b->cfg_host[i].task_joint;
      if ((tj == WBC_JOINT_MANI || tj == WBC_JOINT_HYBRID) && !b->plan_host[i].post_static) all_static = false;
      if (tj != WBC_JOINT_MANI && tj != WBC_JOINT_HYBRID && tj >= WBC_JOINT_MANI) all_static = false;
    }
    if (all_static && !a.in.q_con) { a.post_static = 1; return WBC_OK; }
  }
  if (!b->d_pu) {
    HIP_TRY(hipMalloc((void**)&b->d_pu, sizeof(double) * WBC_V_STRIDE * (size_t)b->max_batch));
    HIP_TRY(hipMalloc((void**)&b->d_pq, sizeof(double) * WBC_Q_STRIDE * (size_t)b->max_batch));
  }
  if (int rc = run_posture(b, B, a.in.q, a.in.model_id, b->d_pu, b->d_pq, stream)) return rc;
  a.in.posture_u = b->d_pu;
  if (literal && !a.in.q_con) a.in.q_con = b->d_pq;
  return WBC_OK;
}

static void fill_args(KernelArgs& a, const WbcBatch* b, int B, double dt) {
  memset(&a, 0, sizeof a);
  a.models = b->d_models; a.cfgs = b->d_cfgs; a.plans = b->d_plans; a.n_models = b->n_models;
  // J'J on the matrix cores: forced (1), off (0) or, by default (-1), for wide Cartesian stacks only — measured on MI355X
  // (profiles/r02_mfma_evidence.txt): +9 % ticks/s at 33 and 45 Cartesian rows (config 2, "everything"), a wash at 6 (config 3)
  a.B = B; a.mrows = b->mrows; a.prows = b->prows; a.mcart = b->mcart; a.jtj_mfma = b->jtj_mfma < 0 ? (b->mcart >= WBC_MFMA_AUTO_ROWS) : b->jtj_mfma; a.presolve = b->presolve; a.presolve_orth = b->presolve_orth ? 1 + any_orth_plan(b) : 0; a.orth_qr = b->orth_qr; a.refine = b->refine; a.sing_tol = b->sing_tol; a.dbg_alias = b->dbg_alias; a.dt = dt;
  a.prof = b->d_prof; a.dbg_stop = b->dbg_stop;
  a.fk_nj = b->max_nj; a.fk_nf = b->max_nf;
}

// The fused tick on the best kernel for the batch: wbc_tick_sim3_kernel (compact LDS, reduced QP only) when every
// model's plan is enabled and the problem fits its layout, followed by wbc_tick_deferred_kernel (general path) over
// the instances it deferred (singular leg block); otherwise the general kernel alone. `a` holds device pointers.
static int launch_update_auto(WbcBatch* b, UpdateArgs& u, int B, void* stream) {
  bool packed = b->packed_update != 0;
  for (int i = 0; i < b->n_models && packed; ++i) packed = b->configured[i] && b->plan_host[i].pk_update_ok != 0;
  u.plans = b->d_plans;
  b->last_update_packed = packed;
  return packed ? launch_update_packed(u, stream) : launch_update(u, B, stream);
}
static bool packed_eligible(const WbcBatch* b, const KernelArgs& a) {
  const bool qcon = a.in.q_con || a.in.posture_u;     // the QCON variant (second kinematics pass at q_con, posture target from posture_u)
  bool packed = b->packed_kernel && a.B >= b->packed_min_batch &&
                !b->count_pivoted && !(b->dbg_stop > 0 && b->dbg_stop < 100) && !b->dbg_alias;   // (dbg_stop 101.. cuts the packed kernel)
  for (int i = 0; i < b->n_models && packed; ++i) packed = qcon ? (b->plan_host[i].packed_ok_pu != 0) : (b->plan_host[i].packed_ok != 0);
  return packed;
}
static bool sim3_eligible(const WbcBatch* b, const KernelArgs& a) {
  if (!b->sim3_kernel || !b->presolve || b->n_models < 1) return false;
  if (b->jtj_mfma > 0) return false;   // forced: the compact kernel has no matrix-core contraction, the option selects the general kernel
  if (a.in.com_target || a.in.com_target_vel) return false;
  if (a.in.ee_ref_rot && !packed_eligible(b, a)) return false;   // orientation references: the packed kernel honours the gripper's, the one-instance compact kernel none
  if (b->prows > WBC_SIM3_MAXP || b->mcart > 12) return false;
  for (int i = 0; i < b->n_models; ++i) {
    const DevPlan& P = b->plan_host[i];
    if (!P.enabled || P.p_keep + (b->cfg_host[i].use_bounds ? 3 * P.nelim : 0) > WBC_SIM3_MAXP) return false;
  }
  return true;
}
// the packed orth kernel: every plan q_ok, nothing passed that it does not read (caller's posture target / constraint state; working sets are accepted: no inequality to seed, an empty set out),
// the orthonormal presolve on, no forced matrix-core contraction (the EE tasks' orientation references are honoured)
// packed_orth: 1 (default) from WBC_ORTHP_MIN_BATCH instances on — below that a launch is a single round of waves and the one-instance
// kernel's shorter dependent chain wins (measured on MI355X, tools/debug_orthp.py: 25 us vs 55 us at B = 1024, equal at 4096, 0.075 vs
// 0.108 ms at 8192, 0.318 vs 0.782 ms at 65536); 2: always (tests)
constexpr int WBC_ORTHP_MIN_BATCH = 4608;
static bool orthp_eligible(const WbcBatch* b, const KernelArgs& a) {
  if (b->packed_orth == 1 && a.B < WBC_ORTHP_MIN_BATCH) return false;
  if (!b->packed_orth || !b->packed_kernel || !b->presolve || !b->presolve_orth || b->n_models < 1 || b->jtj_mfma > 0) return false;
  if (a.in.q_con || a.in.posture_u || b->dbg_alias || (b->dbg_stop > 0 && b->dbg_stop < 200)) return false;   // (dbg_stop 201.. cuts this kernel; working sets: nothing to seed, an empty set out)
  for (int i = 0; i < b->n_models; ++i) if (!b->plan_host[i].q_ok || b->plan_host[i].q_ok != b->plan_host[0].q_ok) return false;
  return true;
}
// the packed box kernel: every plan x_ok, nothing passed that it does not read; packed_box: 1 (default) from WBC_BOXP_MIN_BATCH instances on, 2: always.
// Unlike the packed orth kernel it wins at every batch size (measured on MI355X, tools/small_batch_boxp.py: 28.1 vs 29.3 us at B = 1, 39.9 vs 46.4 us
// at 1024, 46 vs 104 us at 4096): the one-instance kernel's chain through a 23-wide Cholesky and ~5 working-set changes is the longer one.
constexpr int WBC_BOXP_MIN_BATCH = 1;
static bool boxp_eligible(const WbcBatch* b, const KernelArgs& a) {
  if (b->packed_box == 1 && a.B < WBC_BOXP_MIN_BATCH) return false;
  if (!b->packed_box || !b->packed_kernel || !b->presolve || b->n_models < 1 || b->jtj_mfma > 0 || b->prows != 0) return false;
  if (a.in.q_con || a.in.posture_u || b->dbg_alias || (b->dbg_stop > 0 && b->dbg_stop < 300)) return false;   // (dbg_stop 301.. cuts this kernel; working sets: its WARM variant)
  for (int i = 0; i < b->n_models; ++i) if (!b->plan_host[i].x_ok) return false;
  return true;
}
static int launch_tick_auto(WbcBatch* b, KernelArgs& a, int B, void* stream) {
  if (boxp_eligible(b, a)) {    // ONE kernel per tick; last_path 4
    b->last_path = 4;
    b->last_orth = 0;
    if (!a.out.status) {
      if (!b->d_status) HIP_TRY(hipMalloc((void**)&b->d_status, sizeof(int32_t) * (size_t)b->max_batch));
      a.out.status = b->d_status;
    }
    if (!b->d_dstat) {
      HIP_TRY(hipMalloc((void**)&b->d_dstat, sizeof(unsigned long long)));
      HIP_TRY(hipMemsetAsync(b->d_dstat, 0, sizeof(unsigned long long), (hipStream_t)stream));
    }
    a.defer_stat = b->d_dstat;
    a.tick_seq = ++b->tick_seq;
    if (!b->tick_seq) a.tick_seq = ++b->tick_seq;
    if (int e = launch_tick_boxp(a, stream)) return fail(WBC_E_HIP, "packed box tick kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    return WBC_OK;
  }
  if (orthp_eligible(b, a)) {   // ONE kernel per tick; last_path 3
    b->last_path = 3;
    b->last_orth = 1;
    if (!a.out.status) {
      if (!b->d_status) HIP_TRY(hipMalloc((void**)&b->d_status, sizeof(int32_t) * (size_t)b->max_batch));
      a.out.status = b->d_status;
    }
    if (!b->d_dstat) {
      HIP_TRY(hipMalloc((void**)&b->d_dstat, sizeof(unsigned long long)));
      HIP_TRY(hipMemsetAsync(b->d_dstat, 0, sizeof(unsigned long long), (hipStream_t)stream));
    }
    a.defer_stat = b->d_dstat;
    a.tick_seq = ++b->tick_seq;
    if (!b->tick_seq) a.tick_seq = ++b->tick_seq;
    if (int e = launch_tick_orthp(a, stream, b->plan_host[0].q_ok == 2)) return fail(WBC_E_HIP, "packed orth tick kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    return WBC_OK;
  }
  // The one-instance compact kernel (path 1) does not refine: it lives on 168 VGPRs / 13.2 KB LDS (3 waves per SIMD) and has room neither for the
  // task image nor for the Jacobian columns the residual is formed from. With the refinement on (default) what the packed kernel does not take
  // runs on the general kernel, whose structural presolve solves the same reduced problem and refines it (option refine = 0 brings path 1 back).
  if (!sim3_eligible(b, a) || (b->refine > 0 && !packed_eligible(b, a))) {
    b->last_path = 0;
    b->last_orth = a.presolve && a.presolve_orth == 2 && !(a.ws_in || a.ws_out);
    if (int e = launch_tick(a, MODE_TICK, grid_tick(b, B), stream)) return fail(WBC_E_HIP, "tick kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    return WBC_OK;
  }
  b->last_path = 1;
  if (!a.out.status) {
    if (!b->d_status) HIP_TRY(hipMalloc((void**)&b->d_status, sizeof(int32_t) * (size_t)b->max_batch));
    a.out.status = b->d_status;
  }
  if (!b->d_defer) {
    HIP_TRY(hipMalloc((void**)&b->d_defer, sizeof(int32_t) * ((size_t)b->max_batch + 4)));
    // (on the CALL's stream: a null-stream memset is not ordered against a non-blocking caller stream)
    HIP_TRY(hipMemsetAsync(b->d_defer, 0, sizeof(int32_t) * ((size_t)b->max_batch + 4), (hipStream_t)stream));
  }
  a.defer = b->d_defer;                                  // (count = 0 here: wbc_tick_deferred_kernel leaves the list empty behind it)
  a.defer_aux = b->d_defer + 1 + b->max_batch;
  a.dbg_force_defer = b->force_defer;
  if (b->count_pivoted) {
    a.pivot_count = b->d_defer + 1 + b->max_batch;
    HIP_TRY(hipMemsetAsync(a.pivot_count, 0, sizeof(int32_t), (hipStream_t)stream));
  }
  if (packed_eligible(b, a)) {   // ONE kernel per tick: what the packed kernel cannot reduce its own wave redoes on the general path
    b->last_path = 2;
    if (!b->d_dstat) {
      HIP_TRY(hipMalloc((void**)&b->d_dstat, sizeof(unsigned long long)));
      HIP_TRY(hipMemsetAsync(b->d_dstat, 0, sizeof(unsigned long long), (hipStream_t)stream));
    }
    a.defer_stat = b->d_dstat;
    a.packed_trunk = b->cfg_host[0].task_trunk != 0;
    a.tick_seq = ++b->tick_seq;
    if (!b->tick_seq) a.tick_seq = ++b->tick_seq;          // (0 is the cleared word's sequence number)
    if (int e = launch_tick_sim3p(a, stream)) return fail(WBC_E_HIP, "packed sim3 tick kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    return WBC_OK;
  }
  if (int e = launch_tick_sim3(a, B, stream)) return fail(WBC_E_HIP, "sim3 tick kernel launch failed: %s", hipGetErrorString((hipError_t)e));
  if (int e = launch_tick_deferred(a, stream)) {
    // the list the sim3 kernel may have filled stays behind: empty it, or the next tick appends after a stale count
    (void)hipMemsetAsync(b->d_defer, 0, sizeof(int32_t), (hipStream_t)stream);
    return fail(WBC_E_HIP, "deferred tick kernel launch failed: %s", hipGetErrorString((hipError_t)e));
  }
  return WBC_OK;
}

// ---------------------------------------------------------------------------------------------- entry points
extern "C" int wbc_fk_jacobians(WbcBatch* b, int B, const double* q, const int32_t* model_id, int mem,
                                const WbcFkOut* out, void* stream) {
  int rc = check_batch(b, B, "wbc_fk_jacobians", false);
  if (rc) return rc;
  if (!q || !out) return fail(WBC_E_ARG, "wbc_fk_jacobians: null argument");
  if (b->n_models > 1 && !model_id) return fail(WBC_E_ARG, "wbc_fk_jacobians: model_id is required with %d models", b->n_models);
  HIP_TRY(hipSetDevice(b->device_id));
  KernelArgs a;
  fill_args(a, b, B, 0.0);
  if (!b->configured[0]) {   // FK needs no settings; give the kernel a zeroed config to read its (unused) switches from
    WbcConfig z;
    memset(&z, 0, sizeof z);
    for (int i = 0; i < b->n_models; ++i)
      if (!b->configured[i]) HIP_TRY(hipMemcpy(b->d_cfgs + i, &z, sizeof z, hipMemcpyHostToDevice));
  }
  a.in.q = q; a.in.model_id = model_id; a.fk = *out;
  const int nj = b->max_nj, nf = b->max_nf;   // output strides: the largest model of the handle
  Stager st{b, mem, (hipStream_t)stream, {}};
  st.in(&a.in.q, (size_t)B * WBC_Q_STRIDE); st.in(&a.in.model_id, (size_t)B);
  st.out(&a.fk.oMi, (size_t)B * nj * 12); st.out(&a.fk.oMf, (size_t)B * nf * 12);
  st.out(&a.fk.J, (size_t)B * 6 * WBC_V_STRIDE); st.out(&a.fk.com, (size_t)B * 3); st.out(&a.fk.Jcom, (size_t)B * 3 * WBC_V_STRIDE);
  if ((rc = st.stage())) return rc;
  if (int e = launch_tick(a, MODE_FK, grid_tick(b, B), stream)) return fail(WBC_E_HIP, "fk kernel launch failed: %s", hipGetErrorString((hipError_t)e));
  return st.finish();
}

extern "C" int wbc_assemble(WbcBatch* b, int B, const WbcTickIn* in, double dt, int mem, const WbcQpData* out, void* stream) {
  int rc = check_batch(b, B, "wbc_assemble", true);
  if (rc) return rc;
  if (!out || !(dt > 0)) return fail(WBC_E_ARG, "wbc_assemble: null output or dt <= 0");
  if ((rc = validate_tick_in(b, in, "wbc_assemble"))) return rc;
  HIP_TRY(hipSetDevice(b->device_id));
  KernelArgs a;
  fill_args(a, b, B, dt);
  a.in = *in; a.qp = *out;
  Stager st{b, mem, (hipStream_t)stream, {}};
  stage_tick_in(st, a.in, B, b);
  const size_t n = (size_t)B, m = b->mrows, p = b->prows, V = WBC_V_STRIDE;
  st.out(&a.qp.A, n * m * V); st.out(&a.qp.b, n * m); st.out(&a.qp.H, n * V * V); st.out(&a.qp.g, n * V);
  st.out(&a.qp.C, n * p * V); st.out(&a.qp.Clb, n * p); st.out(&a.qp.Cub, n * p); st.out(&a.qp.lb, n * V); st.out(&a.qp.ub, n * V);
  if ((rc = st.stage())) return rc;
  if ((rc = auto_posture(b, a, B, stream))) return rc;
  if (int e = launch_tick(a, MODE_ASSEMBLE, grid_tick(b, B), stream)) return fail(WBC_E_HIP, "assemble kernel launch failed: %s", hipGetErrorString((hipError_t)e));
  return st.finish();
}

extern "C" int wbc_tick(WbcBatch* b, int B, const WbcTickIn* in, double dt, int mem, const WbcTickOut* out, void* stream) {
  int rc = check_batch(b, B, "wbc_tick", true);
  if (rc) return rc;
  if (!out || !out->qdot || !(dt > 0)) return fail(WBC_E_ARG, "wbc_tick: qdot output required and dt > 0");
  if ((rc = validate_tick_in(b, in, "wbc_tick"))) return rc;
  HIP_TRY(hipSetDevice(b->device_id));
  KernelArgs a;
  fill_args(a, b, B, dt);
  a.in = *in; a.out = *out;
  Stager st{b, mem, (hipStream_t)stream, {}};
  stage_tick_in(st, a.in, B, b);
  const size_t n = (size_t)B;
  st.out(&a.out.qdot, n * WBC_V_STRIDE); st.out(&a.out.status, n); st.out(&a.out.iters, n); st.out(&a.out.q_next, n * WBC_Q_STRIDE);
  st.out(&a.out.working_set, n * 2);
  if ((rc = st.stage())) return rc;
  a.ws_in = (const unsigned long long*)a.in.working_set; a.ws_out = (unsigned long long*)a.out.working_set;
  if ((rc = auto_posture(b, a, B, stream))) return rc;
  if ((rc = launch_tick_auto(b, a, B, stream))) return rc;
  return st.finish();
}

extern "C" int wbc_posture_target(WbcBatch* b, int B, const double* q, const int32_t* model_id, int mem, double* u, double* q_after,
                                  void* stream) {
  int rc = check_batch(b, B, "wbc_posture_target", true);
  if (rc) return rc;
  if (!q || !u) return fail(WBC_E_ARG, "wbc_posture_target: q and u required");
  if (b->n_models > 1 && !model_id) return fail(WBC_E_ARG, "wbc_posture_target: model_id is required with %d models", b->n_models);
  HIP_TRY(hipSetDevice(b->device_id));
  Stager st{b, mem, (hipStream_t)stream, {}};
  st.in(&q, (size_t)B * WBC_Q_STRIDE); st.in(&model_id, (size_t)B);
  st.out(&u, (size_t)B * WBC_V_STRIDE); st.out(&q_after, (size_t)B * WBC_Q_STRIDE);
  if ((rc = st.stage())) return rc;
  if ((rc = run_posture(b, B, q, model_id, u, q_after, stream))) return rc;
  return st.finish();
}

extern "C" int wbc_update_state(WbcBatch* b, int B, const double* q_cur, const double* q_next, const double* imu,
                                const double* foot_targets, const int32_t* model_id, int mem, double* q_new, void* stream) {
  int rc = check_batch(b, B, "wbc_update_state", true);
  if (rc) return rc;
  if (!q_cur || !q_next || !foot_targets || !q_new) return fail(WBC_E_ARG, "wbc_update_state: q_cur, q_next, foot_targets and q_new are required");
  if (b->n_models > 1 && !model_id) return fail(WBC_E_ARG, "wbc_update_state: model_id is required with %d models", b->n_models);
  HIP_TRY(hipSetDevice(b->device_id));
  UpdateArgs a;
  memset(&a, 0, sizeof a);
  a.models = b->d_models; a.cfgs = b->d_cfgs; a.B = B; a.n_models = b->n_models;
  a.q_cur = q_cur; a.q_next = q_next; a.imu = imu; a.foot_targets = foot_targets; a.model_id = model_id; a.q_new = q_new;
  Stager st{b, mem, (hipStream_t)stream, {}};
  const size_t n = (size_t)B;
  const bool alias = (const double*)q_new == q_cur;
  st.in(&a.q_cur, n * WBC_Q_STRIDE); st.in(&a.q_next, n * WBC_Q_STRIDE); st.in(&a.imu, n * 4); st.in(&a.foot_targets, n * 15);
  st.in(&a.model_id, n);
  st.out(&a.q_new, n * WBC_Q_STRIDE);
  if ((rc = st.stage())) return rc;
  (void)alias;
  if (int e = launch_update_auto(b, a, B, stream)) return fail(WBC_E_HIP, "update kernel launch failed: %s", hipGetErrorString((hipError_t)e));
  return st.finish();
}

// K closed-loop ticks: the mutable controller state lives in the handle's rollout workspace; in0 is only read.
extern "C" int wbc_rollout(WbcBatch* b, int B, const WbcTickIn* in0, double dt, const WbcRollout* r, int mem, void* stream) {
  int rc = check_batch(b, B, "wbc_rollout", true);
  if (rc) return rc;
  if (!r || r->ticks < 1 || !(dt > 0)) return fail(WBC_E_ARG, "wbc_rollout: ticks >= 1 and dt > 0 required");
  if (r->hold_ticks < 0 || (r->mode != WBC_ROLLOUT_RUNNING && r->mode != WBC_ROLLOUT_WARMUP))
    return fail(WBC_E_ARG, "wbc_rollout: hold_ticks >= 0 and mode WBC_ROLLOUT_RUNNING / WBC_ROLLOUT_WARMUP required");
  if ((rc = validate_tick_in(b, in0, "wbc_rollout"))) return rc;
  if (!in0->ee_target || !in0->prev_ee_target) return fail(WBC_E_ARG, "wbc_rollout: ee_target / prev_ee_target are required (the base estimator reads the foot targets)");
  HIP_TRY(hipSetDevice(b->device_id));
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)B, NB = (size_t)b->max_batch;
  // workspace layout (doubles per instance), then two int32 per instance
  enum { O_Q = 0, O_QN = 27, O_QD = 54, O_EET = 80, O_EEP = 95, O_TT = 110, O_TP = 113, O_EPR = 116, O_TPR = 161, O_END = 170 };
  if (!b->d_roll) HIP_TRY(hipMalloc(&b->d_roll, NB * (O_END * sizeof(double) + 2 * sizeof(int32_t) + 2 * sizeof(unsigned long long))));
  double* W = (double*)b->d_roll;
  auto blk = [&](int off) { return W + NB * (size_t)off; };
  int32_t* w_status = (int32_t*)(W + NB * O_END);
  int32_t* w_iters = w_status + NB;
  unsigned long long* w_ws = (unsigned long long*)(w_iters + NB);     // [NB][2] working sets carried from tick to tick (8-byte aligned)

  KernelArgs a;
  fill_args(a, b, B, dt);
  a.in = *in0;
  WbcRollout ro = *r;
  Stager st{b, mem, s, {}};
  stage_tick_in(st, a.in, B, b);
  st.in(&ro.ee_target_step, n * 15); st.in(&ro.trunk_target_step, n * 3); st.in(&ro.imu, n * 4);
  st.out(&ro.q_final, n * WBC_Q_STRIDE); st.out(&ro.qdot_last, n * WBC_V_STRIDE); st.out(&ro.ee_target_final, n * 15);
  st.out(&ro.grip_trace, (size_t)(r->ticks + r->hold_ticks) * n * 3); st.out(&ro.status_max, n); st.out(&ro.iters_sum, n);
  if ((rc = st.stage())) return rc;
  // seed the mutable state from in0
  auto seed = [&](int off, const double* src, size_t k) -> int {
    if (src) HIP_TRY(hipMemcpyAsync(blk(off), src, n * k * sizeof(double), hipMemcpyDeviceToDevice, s));
    return WBC_OK;
  };
  if ((rc = seed(O_Q, a.in.q, 27)) || (rc = seed(O_EET, a.in.ee_target, 15)) || (rc = seed(O_EEP, a.in.prev_ee_target, 15)) ||
      (rc = seed(O_TT, a.in.trunk_target, 3)) || (rc = seed(O_TP, a.in.prev_trunk_target, 3)) ||
      (rc = seed(O_EPR, a.in.ee_prev_rot, 45)) || (rc = seed(O_TPR, a.in.trunk_prev_rot, 9))) return rc;
  if (b->warm_start) {   // first tick: the caller's working set if there is one, else cold
    if (a.in.working_set) HIP_TRY(hipMemcpyAsync(w_ws, a.in.working_set, n * 2 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
    else HIP_TRY(hipMemsetAsync(w_ws, 0, n * 2 * sizeof(unsigned long long), s));
    a.ws_in = w_ws; a.ws_out = w_ws;
  }
  if (ro.status_max) HIP_TRY(hipMemsetAsync(ro.status_max, 0, n * sizeof(int32_t), s));
  if (ro.iters_sum) HIP_TRY(hipMemsetAsync(ro.iters_sum, 0, n * sizeof(int32_t), s));
  const WbcTickIn first = a.in;
  a.in.q = blk(O_Q); a.in.ee_target = blk(O_EET); a.in.prev_ee_target = blk(O_EEP);
  if (first.trunk_target) a.in.trunk_target = blk(O_TT);
  if (first.prev_trunk_target) a.in.prev_trunk_target = blk(O_TP);
  if (first.ee_prev_rot) a.in.ee_prev_rot = blk(O_EPR);
  if (first.trunk_prev_rot) a.in.trunk_prev_rot = blk(O_TPR);
  a.out.qdot = blk(O_QD); a.out.q_next = blk(O_QN); a.out.status = w_status; a.out.iters = w_iters;

  UpdateArgs u;
  memset(&u, 0, sizeof u);
  u.models = b->d_models; u.cfgs = b->d_cfgs; u.B = B; u.n_models = b->n_models; u.mode = r->mode;
  u.q_cur = blk(O_Q); u.q_next = blk(O_QN); u.imu = ro.imu; u.foot_targets = blk(O_EET); u.model_id = a.in.model_id; u.q_new = blk(O_Q);
  u.ee_target = blk(O_EET); u.prev_ee_target = blk(O_EEP);
  u.trunk_target = first.trunk_target ? blk(O_TT) : nullptr; u.prev_trunk_target = first.prev_trunk_target ? blk(O_TP) : nullptr;
  u.ee_prev_rot = first.ee_prev_rot ? blk(O_EPR) : nullptr; u.trunk_prev_rot = first.trunk_prev_rot ? blk(O_TPR) : nullptr;
  u.ee_ref_rot = first.ee_ref_rot; u.trunk_ref_euler = first.trunk_ref_euler;
  u.ee_step = ro.ee_target_step; u.trunk_step = ro.trunk_target_step;
  u.status = w_status; u.iters = w_iters; u.status_max = ro.status_max; u.iters_sum = ro.iters_sum;
  const WbcTickIn loop_in = a.in;
  for (int k = 0; k < r->ticks + r->hold_ticks; ++k) {
    if (k == r->ticks) { u.ee_step = nullptr; u.trunk_step = nullptr; }   // hold phase: the targets stay where they are
    a.in = loop_in;                       // auto_posture fills posture_u / q_con afresh every tick
    if ((rc = auto_posture(b, a, B, stream))) return rc;
    if ((rc = launch_tick_auto(b, a, B, stream))) return rc;
    u.grip_trace = ro.grip_trace ? ro.grip_trace + (size_t)k * n * 3 : nullptr;
    if (int e = launch_update_auto(b, u, B, stream)) return fail(WBC_E_HIP, "update kernel launch failed: %s", hipGetErrorString((hipError_t)e));
  }
  if (ro.q_final) HIP_TRY(hipMemcpyAsync(ro.q_final, blk(O_Q), n * 27 * sizeof(double), hipMemcpyDeviceToDevice, s));
  if (ro.qdot_last) HIP_TRY(hipMemcpyAsync(ro.qdot_last, blk(O_QD), n * 26 * sizeof(double), hipMemcpyDeviceToDevice, s));
  if (ro.ee_target_final) HIP_TRY(hipMemcpyAsync(ro.ee_target_final, blk(O_EET), n * 15 * sizeof(double), hipMemcpyDeviceToDevice, s));
  return st.finish();
}

static int qp_common(WbcBatch* b, int B, QpArgs& a, int mem, void* stream, const char* who) {
  int rc = check_batch(b, B, who, false, false);
  if (rc) return rc;
  const int n = a.n, p = a.p, m = a.m;
  if (n < 1 || n > WBC_MAX_NV || p < 0 || p > WBC_MAX_P || m < 0 || m > WBC_MAX_M) return fail(WBC_E_ARG, "%s: n = %d, p = %d, m = %d out of range", who, n, p, m);
  if (!a.x) return fail(WBC_E_ARG, "%s: x output required", who);
  if (p > 0 && (!a.C || !a.Clb || !a.Cub)) return fail(WBC_E_ARG, "%s: C, Clb, Cub required when p > 0", who);
  if ((a.lb != nullptr) != (a.ub != nullptr)) return fail(WBC_E_ARG, "%s: lb and ub go together", who);
  HIP_TRY(hipSetDevice(b->device_id));
  Stager st{b, mem, (hipStream_t)stream, {}};
  const size_t N = (size_t)B;
  st.in(&a.H, N * n * n); st.in(&a.g, N * n); st.in(&a.A, N * m * n); st.in(&a.bvec, N * m);
  st.in(&a.C, N * p * n); st.in(&a.lb, N * n); st.in(&a.ub, N * n); st.in(&a.Clb, N * p); st.in(&a.Cub, N * p);
  st.out(&a.x, N * n); st.out(&a.status, N); st.out(&a.iters, N); st.out(&a.H_out, N * n * n); st.out(&a.g_out, N * n);
  st.in(&a.ws_in, N * 2); st.out(&a.ws_out, N * 2);   // (host buffers are staged separately, so in and out may be the same host array)
  if ((rc = st.stage())) return rc;
  a.refine = b->refine; a.dbg_stop = b->dbg_stop;
  // several problems per wavefront (wbc_k_qpp.hip) unless the call carries working sets (hot start) or option packed_kernel is 0
  const int lanes = b->packed_kernel ? qp_packed_lanes(a) : 0;
  b->last_qp_path = lanes ? lanes : 1;
  if (int e = lanes ? launch_qp_packed(a, stream) : launch_qp(a, grid_for(b, B), stream)) return fail(WBC_E_HIP, "qp kernel launch failed: %s", hipGetErrorString((hipError_t)e));
  return st.finish();
}

extern "C" int wbc_qp_solve(WbcBatch* b, int B, int n, int p, const double* H, const double* g, const double* C,
                            const double* lb, const double* ub, const double* Clb, const double* Cub, int mem,
                            double* x, int32_t* status, int32_t* iters, const uint64_t* working_set_in, uint64_t* working_set_out,
                            void* stream) {
  if (!H || !g) return fail(WBC_E_ARG, "wbc_qp_solve: H and g required");
  QpArgs a;
  memset(&a, 0, sizeof a);
  a.B = B; a.n = n; a.p = p; a.m = 0;
  a.H = H; a.g = g; a.C = C; a.lb = lb; a.ub = ub; a.Clb = Clb; a.Cub = Cub; a.x = x; a.status = status; a.iters = iters;
  a.ws_in = (const unsigned long long*)working_set_in; a.ws_out = (unsigned long long*)working_set_out;
  return qp_common(b, B, a, mem, stream, "wbc_qp_solve");
}

extern "C" int wbc_qp_solve_ls(WbcBatch* b, int B, int m, int n, int p, const double* A, const double* bvec, const double* C,
                               const double* lb, const double* ub, const double* Clb, const double* Cub, int mem, int use_mfma,
                               double* x, int32_t* status, int32_t* iters, double* H_out, double* g_out,
                               const uint64_t* working_set_in, uint64_t* working_set_out, void* stream) {
  if (!A || !bvec || m < 1) return fail(WBC_E_ARG, "wbc_qp_solve_ls: A, b and m >= 1 required");
  QpArgs a;
  memset(&a, 0, sizeof a);
  a.B = B; a.n = n; a.p = p; a.m = m; a.use_mfma = use_mfma < 0 ? (m >= WBC_MFMA_AUTO_ROWS) : (use_mfma != 0);
  a.A = A; a.bvec = bvec; a.C = C; a.lb = lb; a.ub = ub; a.Clb = Clb; a.Cub = Cub;
  a.x = x; a.status = status; a.iters = iters; a.H_out = H_out; a.g_out = g_out;
  a.ws_in = (const unsigned long long*)working_set_in; a.ws_out = (unsigned long long*)working_set_out;
  return qp_common(b, B, a, mem, stream, "wbc_qp_solve_ls");
}

extern "C" int wbc_integrate(WbcBatch* b, int B, const double* q, const double* v, const int32_t* model_id, double dt, int mem,
                             double* q_next, void* stream) {
  int rc = check_batch(b, B, "wbc_integrate", false);
  if (rc) return rc;
  if (!q || !v || !q_next) return fail(WBC_E_ARG, "wbc_integrate: null argument");
  if (b->n_models > 1 && !model_id) return fail(WBC_E_ARG, "wbc_integrate: model_id is required with %d models", b->n_models);
  HIP_TRY(hipSetDevice(b->device_id));
  IntegrateArgs a;
  memset(&a, 0, sizeof a);
  a.models = b->d_models; a.B = B; a.n_models = b->n_models; a.dt = dt; a.q = q; a.v = v; a.model_id = model_id; a.q_next = q_next;
  Stager st{b, mem, (hipStream_t)stream, {}};
  st.in(&a.q, (size_t)B * WBC_Q_STRIDE); st.in(&a.v, (size_t)B * WBC_V_STRIDE); st.in(&a.model_id, (size_t)B);
  st.out(&a.q_next, (size_t)B * WBC_Q_STRIDE);
  if ((rc = st.stage())) return rc;
  if (int e = launch_integrate(a, grid_for(b, B), stream)) return fail(WBC_E_HIP, "integrate kernel launch failed: %s", hipGetErrorString((hipError_t)e));
  return st.finish();
}

extern "C" int wbc_debug_cycles(WbcBatch* b, uint64_t* out24) {
  if (!b || !out24) return fail(WBC_E_ARG, "wbc_debug_cycles: null argument");
  HIP_TRY(hipSetDevice(b->device_id));
  if (!b->d_prof) {   // first call arms the counters (they stay zero in non-profile builds)
    HIP_TRY(hipMalloc((void**)&b->d_prof, 24 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(b->d_prof, 0, 24 * sizeof(unsigned long long)));
  }
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out24, b->d_prof, 24 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(b->d_prof, 0, 24 * sizeof(unsigned long long)));
  return WBC_OK;
}

extern "C" const char* wbc_last_error(void) { return g_err; }
#ifdef WBC_PROFILE
extern "C" const char* wbc_version(void) { return "wbc-hip 0.1 (gfx950, PROFILE build)"; }
#else
extern "C" const char* wbc_version(void) { return "wbc-hip 0.1 (gfx950)"; }
#endif
extern "C" int wbc_abi_sizes(int32_t* sb, int32_t* sc) {
  if (sb) *sb = (int32_t)sizeof(WbcModelBlob);
  if (sc) *sc = (int32_t)sizeof(WbcConfig);
  return WBC_OK;
}
