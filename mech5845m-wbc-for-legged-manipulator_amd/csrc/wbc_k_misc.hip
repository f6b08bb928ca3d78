// wbc_k_misc.hip — stand-alone QP, integrate, posture-target and state-update kernels.
#include "wbc_packed.h"

namespace wbc {

// Stand-alone QP (QP_Wrapper.QP.solveQP): H, g (or A, b) and constraints straight from HBM.
// NM: compiled size of the dual active-set core (launch_qp picks the smallest of 12 / 16 / 24 / 26 that holds n: its sweeps cost ~NM^2)
// WARM: working sets in / out (QP.solveQPHotstart, QP_Wrapper.py:55-73): [B][2] words in the problem's own indexing — word 0: bit i / 32 + i =
// variable i at its lower / upper bound, word 1: constraint row i
template <int NM, bool WARM = false>
__global__ void __launch_bounds__(64, 2) wbc_qp_kernel(const QpArgs A) {
  __shared__ Smem S;
  S.cl[threadIdx.x] = 0.0;
#pragma unroll 1
  for (int b = blockIdx.x; b < A.B; b += gridDim.x) {
    int lane = threadIdx.x, n = A.n, p = A.p, m = A.m;
    asm volatile("" : "+v"(lane), "+s"(n), "+s"(p), "+s"(m));   // no LICM of lane/n-derived masks
    const int li = li_clamp(lane);
    double g = 0.0;
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
        });
        g = (lane < n) ? -S.npv[li] : 0.0;
        if (n < NV) {                                   // padded rows / columns of H are exactly zero off the diagonal
          WSYNC();
        }
      } else {
        // vector path: A is staged by DoF (At[dof][row]) in chunks of <= 48 rows; H accumulates in RA
        double* const At = S.RB;
        constexpr int CH = 48, MT = 50;
        if (lane < NV) for (int k = 0; k < NV; k += 2) sts2(S.RA + lane * LDJ + k, 0.0, 0.0);
#pragma unroll 1
        for (int r0 = 0; r0 < m; r0 += CH) {
          const int mc = (m - r0 < CH) ? m - r0 : CH;
#pragma unroll 1
          for (int r = 0; r < CH; ++r) {
            const double a = (r < mc && lane < n) ? Ab[(size_t)(r0 + r) * n + lane] : 0.0;
            if (lane < NV) At[lane * MT + r] = a;
            if (r < mc) g = fma(-a, bb[r0 + r], g);
          }
          WSYNC();
#pragma unroll 1
          for (int i = 0; i < n; ++i) {
            double s = S.RA[li * LDJ + i];
#pragma unroll 2
            for (int r = 0; r < CH; r += 2) {
              const double2a x2 = lds2(At + i * MT + r); const double2a y2 = lds2(At + li * MT + r);
              s = fma(x2.x, y2.x, fma(x2.y, y2.y, s));
            }
            if (lane < NV) S.RA[lane * LDJ + i] = s;
          }
          WSYNC();
        }
      }
      if (lane >= n && lane < NV) S.RA[lane * LDJ + lane] = 1.0;   // padded DoF
      WSYNC();
      if (A.H_out && lane < n) { double* o = A.H_out + (size_t)b * n * n + (size_t)lane * n; for (int k = 0; k < n; ++k) o[k] = S.RA[lane * LDJ + k]; }
      if (A.g_out && lane < n) A.g_out[(size_t)b * n + lane] = g;
    } else {
      const double* Hb = A.H + (size_t)b * n * n;
#pragma unroll 1
      for (int idx = lane; idx < NV * NV; idx += 64) {
        const int r = idx / NV, c = idx - r * NV;
        S.RA[r * LDJ + c] = (r < n && c < n) ? Hb[(size_t)r * n + c] : ((r == c) ? 1.0 : 0.0);
      }
      g = (lane < n) ? A.g[(size_t)b * n + lane] : 0.0;
    }
#pragma unroll 1
    for (int r = 0; r < p; ++r)
      if (lane < NV) S.RC[r * LDJ + lane] = (lane < n) ? A.C[((size_t)b * p + r) * n + lane] : 0.0;
    const double lb = (lane < n) ? (A.lb ? A.lb[(size_t)b * n + lane] : -1e30) : 0.0;
    const double ub = (lane < n) ? (A.ub ? A.ub[(size_t)b * n + lane] : 1e30) : 0.0;
    const double clb = (lane < p) ? A.Clb[(size_t)b * p + lane] : 0.0;
    const double cub = (lane < p) ? A.Cub[(size_t)b * p + lane] : 0.0;
    WSYNC();
    unsigned long long ts[T_NN];
    (void)ts;
    int sb = 0, sr = 0;
    if (WARM && A.ws_in) {
      const unsigned long long w0 = A.ws_in[2 * (size_t)b], w1 = A.ws_in[2 * (size_t)b + 1];
      sb = (lane < 32) ? (int)(((w0 >> lane) & 1ull) | (((w0 >> (32 + lane)) & 1ull) << 1)) : 0;
      sr = (lane < 32) ? (int)(((w1 >> lane) & 1ull) | (((w1 >> (32 + lane)) & 1ull) << 1)) : 0;
    }
    // the refinement's residual from the caller's own data in HBM (L2-resident: just read): QP(A, b, ...) -> A'(b - A x), rows on lanes
    // (lane r: rows r and r + 64), e through S.in (unused here), then column k on lane k; QP(H, g) -> -(H x + g), the only residual it has
    auto resid = [&](const double xk) -> double {
      WSYNC();
      if (lane < 32) S.xv[lane] = (lane < n) ? xk : 0.0;
      WSYNC();
      double r = 0.0;
      if (m > 0) {
        const double* Ab = A.A + (size_t)b * m * n;
        const double* bb = A.bvec + (size_t)b * m;
#pragma unroll 1
        for (int r0 = 0; r0 < m; r0 += 64) {
          const int row = r0 + lane;
          if (row < m) {
            double e = bb[row];
            for (int k = 0; k < n; ++k) e = fma(-Ab[(size_t)row * n + k], S.xv[k], e);
            S.in[row] = e;
          }
        }
        WSYNC();
        if (lane < n) for (int row = 0; row < m; ++row) r = fma(Ab[(size_t)row * n + lane], S.in[row], r);
      } else if (lane < n) {
        const double* Hb = A.H + (size_t)b * n * n + (size_t)lane * n;
        double t_ = A.g[(size_t)b * n + lane];
        for (int k = 0; k < n; ++k) t_ = fma(Hb[k], S.xv[k], t_);
        r = -t_;
      }
      WSYNC();
      return r;
    };
    typedef Refine<decltype(resid)> RF_;
    const RF_ rf{resid};
    const QpResult res = qp_core<NM, Smem, LDJ, WARM, RF_>(S, g, lb, ub, clb, cub, n, p, lane, ts, 0, sb == 3 ? 0 : sb, sr == 3 ? 0 : sr, rf, (m > 0) ? A.refine : 0, true);   // (H, g) alone: the residual
    // could only come from H, whose rounding IS the error — measured: no gain (6.4e-7 stays 6.4e-7) for 23 % of the solve, so the step is skipped
    if (WARM && A.ws_out) {
      const unsigned long long o0 = (__ballot(res.ws_b == 1) & 0xFFFFFFFFull) | (__ballot(res.ws_b == 2) << 32);
      const unsigned long long o1 = (__ballot(res.ws_r == 1) & 0xFFFFFFFFull) | (__ballot(res.ws_r == 2) << 32);
      if (lane == 0) { A.ws_out[2 * (size_t)b] = o0; A.ws_out[2 * (size_t)b + 1] = o1; }
    }
    if (lane < n) A.x[(size_t)b * n + lane] = res.x;
    if (lane == 0) {
      if (A.status) A.status[b] = res.status;
      if (A.iters) A.iters[b] = res.iters;
    }
    WSYNC();
  }
}

// One translation unit per PART (csrc/Makefile compiles this file once per part, in parallel): part 0 holds every other kernel of this file and
// the launchers; parts 1-3 hold the stand-alone QP kernel's size / warm-start variants (each carries a whole qp_core<NM>).
#ifndef MISC_PART
#define MISC_PART -1      // -1: everything in one unit
#endif
#define KINST(...) template __global__ void wbc_qp_kernel<__VA_ARGS__>(const QpArgs);
#define KDECL(...) extern template __global__ void wbc_qp_kernel<__VA_ARGS__>(const QpArgs);
#if MISC_PART == 1 || MISC_PART == -1
KINST(12)
KINST(16)
#elif MISC_PART == 0
KDECL(12)
KDECL(16)
#endif
#if MISC_PART == 2 || MISC_PART == -1
KINST(24)
KINST(NV)
#elif MISC_PART == 0
KDECL(24)
KDECL(NV)
#endif
#if MISC_PART == 3 || MISC_PART == -1
KINST(16, true)
KINST(NV, true)
#elif MISC_PART == 0
KDECL(16, true)
KDECL(NV, true)
#endif
#undef KINST
#undef KDECL
#if MISC_PART <= 0
int launch_qp(const QpArgs& a, int grid, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (a.ws_in || a.ws_out) {   // hot start: the core sizes the tick problems come in (a reduced problem fits 16, the full one needs 26)
    if (a.n <= 16) hipLaunchKernelGGL((wbc_qp_kernel<16, true>), dim3(grid), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((wbc_qp_kernel<NV, true>), dim3(grid), dim3(64), 0, s, a);
  }
  else if (a.n <= 12) hipLaunchKernelGGL(wbc_qp_kernel<12>, dim3(grid), dim3(64), 0, s, a);
  else if (a.n <= 16) hipLaunchKernelGGL(wbc_qp_kernel<16>, dim3(grid), dim3(64), 0, s, a);
  else if (a.n <= 24) hipLaunchKernelGGL(wbc_qp_kernel<24>, dim3(grid), dim3(64), 0, s, a);
  else hipLaunchKernelGGL(wbc_qp_kernel<NV>, dim3(grid), dim3(64), 0, s, a);
  return check_launch("qp");
}

// pin.integrate for a batch (Robot_Wrapper4.py:440-441): q_next = q (+) v * dt
__global__ void __launch_bounds__(64) wbc_integrate_kernel(const IntegrateArgs A) {
  __shared__ Smem S;
#pragma unroll 1
  for (int b = blockIdx.x; b < A.B; b += gridDim.x) {
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));
    const DevModel& M = A.models[model_index(A.model_id, b, A.n_models)];
    const int nv = M.nv, nq = M.nq;
    if (lane < 32) S.in[IN_Q + lane] = (lane < nq) ? A.q[(size_t)b * NQ + lane] : 0.0;
    const double v = (lane < nv) ? A.v[(size_t)b * NV + lane] * A.dt : 0.0;
    if (lane < 32) S.xv[lane] = v;
    WSYNC();
    double* qn = A.q_next + (size_t)b * NQ;
    integrate_ff(S, lane, qn);
    const int cq = M.col_q[lane & 31];
    if (lane >= 6 && lane < nv) qn[cq] = S.in[IN_Q + cq] + v;
    if (lane >= nq && lane < NQ) qn[lane] = 0.0;
    WSYNC();
  }
}

// ------------------------------------------------------------------------------------------------
// qpJointb "MANI" / "HYBRID" (Robot_Wrapper4.py:1220-1260): u_i = (f(q + d e) - f(q - d e)) / (2 d), f = sqrt(det(J J'))
// of pin.getJointJacobian(joint_id, LOCAL_WORLD_ALIGNED); one instance per wave, the 2 x (6 or 26) perturbed
// configurations are evaluated one after the other (each is a full FK: lane j = joint j, then lane k = column k).
// literal (cfg.posture_literal): the reference's index arithmetic and accumulating perturbations.
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) PSmem {
  double oMi[24 * 12];
  double q[32];
  double Jm[32 * 6];     // column k of the 6 x nv joint Jacobian at Jm[6 k ..]
  double G[36];
};
// det of the symmetric positive semi-definite G = J J' by elimination without pivoting (numpy's det pivots; for an SPD
// matrix both are backward stable and agree to rounding). G is wave-uniform in LDS.
__device__ __forceinline__ double det6_spd(const double* G) {
  double m[6][6];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) m[a][b] = G[6 * a + b];
  double det = 1.0;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const double piv = m[c][c];
    det *= piv;
    const double ip = (piv > 0.0) ? 1.0 / piv : 0.0;
#pragma unroll
    for (int r = c + 1; r < 6; ++r) {
      const double f = m[r][c] * ip;
#pragma unroll
      for (int k = c + 1; k < 6; ++k) m[r][k] = fma(-f, m[c][k], m[r][k]);
    }
  }
  return det > 0.0 ? det : 0.0;
}
__device__ __forceinline__ double manipulability(PSmem& P, const DevModel& M, const LaneConst& lc, const int joint_id,
                                                 const int lane) {
  const int nv = M.nv;
  fk_levels(P.oMi, P.q, load_hdr(M), lc, lane);
  double lin[3], ang[3];
  jac_column(P.oMi, lc, lane, nv, lin, ang);
  const double* Pj = P.oMi + 12 * joint_id;
  const double pj[3] = {Pj[9], Pj[10], Pj[11]};
  const bool sup = (lane < nv) && ((lc.subtree >> joint_id) & 1u);   // column's joint is joint_id or one of its ancestors
  double wxp[3];
  cross3(ang, pj, wxp);
  if (lane < 32) {
#pragma unroll
    for (int r = 0; r < 3; ++r) { P.Jm[6 * lane + r] = sup ? lin[r] + wxp[r] : 0.0; P.Jm[6 * lane + 3 + r] = sup ? ang[r] : 0.0; }
  }
  WSYNC();
  if (lane < 36) {
    const int a = lane / 6, b = lane - 6 * a;
    double s = 0.0;
#pragma unroll 2
    for (int k = 0; k < NV; ++k) s = fma(P.Jm[6 * k + a], P.Jm[6 * k + b], s);
    P.G[lane] = s;
  }
  WSYNC();
  const double f = sqrt(det6_spd(P.G));
  WSYNC();
  return f;
}

__global__ void __launch_bounds__(64) wbc_posture_kernel(const PostureArgs A, const DevModel* __restrict__ models,
                                                         const WbcConfig* __restrict__ cfgs) {
  __shared__ PSmem P;
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int mid = model_index(A.model_id, b, A.n_models);
  const DevModel& M = models[mid];
  const WbcConfig& cfg = cfgs[mid];
  const LaneConst lc = load_lane_const(M, cfg, lane);
  const int nv = M.nv, nq = M.nq;
  const double q0 = (lane < nq) ? A.q[(size_t)b * NQ + lane] : 0.0;
  if (lane < 32) P.q[lane] = q0;
  WSYNC();
  const int mode = cfg.task_joint, literal = cfg.posture_literal;
  const double dq = 0.0002;
  double u = 0.0;
  if ((mode == WBC_JOINT_PREV || mode == WBC_JOINT_HYBRID) && lane < nv) u = P.q[lane < 6 ? lane : lane + 1];   // np.delete(q, 6)
  if (mode == WBC_JOINT_MANI || mode == WBC_JOINT_HYBRID) {
#pragma unroll 1
    for (int i = 0; i < nv; ++i) {
      int joint_id;
      if (mode == WBC_JOINT_MANI) joint_id = (i < 6) ? 1 : (literal ? i + 1 - 5 : i - 4);   // :1226-1229
      else { joint_id = i - 6; if (joint_id < cfg.arm_base_id) continue; }                 // :1250-1251
      if (joint_id >= M.njoints) continue;
      const int qi = literal ? i : ((i < 6) ? i : i + 1);                                   // q[i]: the VELOCITY index (:1231, :1252)
      const double keep = P.q[qi];
      WSYNC();
      // Which joint's angle is q[qi]? The LWA Jacobian of joint_id is built from the axes and origins of its PROPER
      // ancestors and its own origin/axis, none of which the FK derives from the angle of joint_id itself or of any joint
      // outside its ancestor chain (the axis column is an exact copy of the parent's, the origin does not involve the
      // angle). For such a perturbation f1 and f2 are computed from bit-identical inputs, so u_i = 0.5 (f1 - f2)/dq = 0
      // exactly — the twelve FK sweeps of the reference's HYBRID indices all fall in this class.
      const unsigned long long own = __ballot(lc.is_joint && lc.q_idx == qi && lane >= 2);
      const int jp = own ? ctz64(own) : 1;                                                 // qi < 7: the free-flyer (affects everything)
      const bool matters = (jp == 1) || (jp != joint_id && ((M.col_subtree[M.idx_v_of[jp]] >> joint_id) & 1u));
      if (lane == 0) P.q[qi] = keep + dq;
      WSYNC();
      double f1 = 0.0, f2 = 0.0;
      if (matters) f1 = manipulability(P, M, lc, joint_id, lane);
      if (lane == 0) P.q[qi] = (keep + dq) - (dq * 2);
      WSYNC();
      if (matters) f2 = manipulability(P, M, lc, joint_id, lane);
      if (lane == i) u = 0.5 * (f1 - f2) / dq;
      if (!literal) { if (lane == 0) P.q[qi] = keep; WSYNC(); }
    }
  }
  WSYNC();
  if (A.u && lane < NV) A.u[(size_t)b * NV + lane] = (lane < nv) ? u : 0.0;
  if (A.q_after && lane < NQ) A.q_after[(size_t)b * NQ + lane] = (lane < nq) ? P.q[lane] : 0.0;
}

// ------------------------------------------------------------------------------------------------
// qpJointb "MANI" / "HYBRID" (Robot_Wrapper4.py:1220-1260) with every finite-difference point on a LANE OF ITS OWN (round 3): one instance per
// wavefront, lane e = 2 k + side evaluates f = sqrt(det(J J')) of sweep k's joint at q + d e_i (side 0) or (q + d e_i) - 2 d e_i (side 1).
// The reference's loop is sequential only in appearance: in literal mode the perturbations accumulate, but the state sweep k
// sees is known up front — q0 with the entries of the earlier sweeps at (q + d) - 2 d (DevPlan.mp_prev) — so all 2 x mp_n (<= 52) points
// are independent. A lane walks the ancestor chain of its joint once in registers (no cross-lane traffic), forms the WORLD-frame Jacobian
// columns on the way (six of the free-flyer + one per chain joint; det(J J') is the same in every frame the columns may be expressed in)
// and accumulates them straight into G = J J' (21 entries), then det by the elimination of wbc_posture_kernel. sin / cos of every joint angle in its three possible states (q, q + d,
// (q + d) - 2 d) are computed once, one per lane, and shared through LDS. Sweeps that cannot change f (DevPlan: not in the list) are u = 0.
// wbc_posture_kernel (52 sequential whole-tree sweeps per instance) stays as the fallback and as the cross-check in the tests.
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) MPSmem {
  double q[32];
  double sc[24 * 6];      // joint j: sin, cos of q, of q + d, of (q + d) - 2 d
  double f[64];
  double uo[32];          // u of the swept DoF
};
__global__ void __launch_bounds__(64) wbc_posture_par_kernel(const PostureArgs A, const DevModel* __restrict__ models,
                                                             const WbcConfig* __restrict__ cfgs, const DevPlan* __restrict__ plans) {
  __shared__ MPSmem S;
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int mid = model_index(A.model_id, b, A.n_models);
  const DevModel& M = models[mid];
  const DevPlan& P = plans[mid];
  const int nv = M.nv, nq = M.nq, nj = M.njoints;
  const double dq = 0.0002;
  const double q0 = (lane < nq) ? A.q[(size_t)b * NQ + lane] : 0.0;
  if (lane < 32) S.q[lane] = q0;
  // this lane's evaluation
  const int k = lane >> 1, side = lane & 1;
  const bool on = k < P.mp_n;
  const int kk = on ? k : 0;
  const int my_i = P.mp_i[kk], my_qi = P.mp_qi[kk];
  const unsigned my_prev = P.mp_prev[kk];
  int chain[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) chain[c] = on ? P.mp_chain[kk][c] : -1;
  WSYNC();
  // sin / cos table: lane t < 3 (nj - 2): joint 2 + t / 3 in state t % 3
  {
    const int j = 2 + lane / 3, st = lane - 3 * (lane / 3);
    if (j < nj) {
      const int jt = M.jtype[j];
      if (jt >= WBC_JT_RX && jt <= WBC_JT_RZ) {
        const double a0 = S.q[M.idx_q[j]];
        const double a = (st == 0) ? a0 : ((st == 1) ? a0 + dq : (a0 + dq) - (dq * 2));
        const SinCos t = sincos_cw(a);
        S.sc[6 * j + 2 * st] = t.s; S.sc[6 * j + 2 * st + 1] = t.c;
      }
    }
  }
  WSYNC();
  // state of configuration entry e for this lane: 0 = q, 1 = q + d, 2 = (q + d) - 2 d
  auto state_of = [&](const int e) -> int { return (e == my_qi) ? (side ? 2 : 1) : (((my_prev >> e) & 1u) ? 2 : 0); };
  auto value_of = [&](const int e) -> double {
    const double a0 = S.q[e];
    const int st = state_of(e);
    return (st == 0) ? a0 : ((st == 1) ? a0 + dq : (a0 + dq) - (dq * 2));
  };
  double f = 0.0;
  {
    // the free-flyer: R from the (possibly perturbed, not renormalised) quaternion exactly as the FK does, p = xyz
    const double qq[4] = {value_of(3), value_of(4), value_of(5), value_of(6)};
    double R1[9];
    quat_to_R(qq, R1);                                  // row-major
    double G[21];
#pragma unroll
    for (int i = 0; i < 21; ++i) G[i] = 0.0;
    auto add_col = [&](const double* c) {               // G += c c' (upper triangle, row-major packed)
      int t = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int bb = a; bb < 6; ++bb) { G[t] = fma(c[a], c[bb], G[t]); ++t; }
    };
    // ONE walk down the chain with the columns expressed at the BASE origin ((p - p_base) x axis, axis): the reference's LOCAL_WORLD_ALIGNED
    // Jacobian at the joint's origin pJ is X J_base with X = [I, -[pJ - p_base]x; 0, I], det X = 1, so det(J J') — all that f is — does not
    // depend on where the columns are expressed, and the joint's origin need not be known before the columns are formed (the first version
    // walked the chain twice for it). Origins relative to the base: the base position drops out of the arithmetic altogether, so its three
    // sweeps give f1 == f2 bit for bit and u = 0 exactly, as the reference's (and the sequential kernel's) LOCAL_WORLD_ALIGNED form does.
    {
      double X[3] = {R1[0], R1[3], R1[6]}, Y[3] = {R1[1], R1[4], R1[7]}, Z[3] = {R1[2], R1[5], R1[8]};   // columns of the parent's rotation
      double p[3] = {0.0, 0.0, 0.0};                    // origins relative to the base
      // free-flyer columns at its own origin: linear DoF (R e_i, 0), angular DoF (0, R e_i)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double* ax = (i == 0) ? X : ((i == 1) ? Y : Z);
        const double cl[6] = {ax[0], ax[1], ax[2], 0.0, 0.0, 0.0};
        add_col(cl);
        const double ca[6] = {0.0, 0.0, 0.0, ax[0], ax[1], ax[2]};
        add_col(ca);
      }
#pragma unroll 1
      for (int c = 0; c < 8; ++c) {
        const int j = chain[c];
        if (j < 0) continue;
        const int a = M.ax0[j], jt = M.jtype[j];
        const bool rev = jt >= WBC_JT_RX && jt <= WBC_JT_RZ;
        const double t0 = M.tp[j][0], t1 = M.tp[j][1], t2 = M.tp[j][2];
        const int qe = M.idx_q[j];
        const int st = state_of(qe);
        const double sn = rev ? S.sc[6 * j + 2 * st] : 0.0, cs = rev ? S.sc[6 * j + 2 * st + 1] : 1.0;
        const double pris = rev ? 0.0 : value_of(qe);
        double Av[3], Bv[3], Cv[3];                     // the axis column of the parent's rotation and its cyclic successors
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          Av[rr] = (a == 0) ? X[rr] : ((a == 1) ? Y[rr] : Z[rr]);
          Bv[rr] = (a == 0) ? Y[rr] : ((a == 1) ? Z[rr] : X[rr]);
          Cv[rr] = (a == 0) ? Z[rr] : ((a == 1) ? X[rr] : Y[rr]);
        }
        double nB[3], nC[3];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          nB[rr] = cs * Bv[rr] + sn * Cv[rr];
          nC[rr] = cs * Cv[rr] - sn * Bv[rr];
          p[rr] = p[rr] + Av[rr] * (t0 + pris) + Bv[rr] * t1 + Cv[rr] * t2;
        }
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          const double na = Av[rr], nb = nB[rr], nc = nC[rr];
          X[rr] = (a == 0) ? na : ((a == 1) ? nc : nb);
          Y[rr] = (a == 0) ? nb : ((a == 1) ? na : nc);
          Z[rr] = (a == 0) ? nc : ((a == 1) ? nb : na);
        }
        double col[6];                                  // this joint's column: revolute (p x axis, axis), prismatic (axis, 0)
        if (rev) {
          double cr[3];
          cross3(p, Av, cr);
          col[0] = cr[0]; col[1] = cr[1]; col[2] = cr[2]; col[3] = Av[0]; col[4] = Av[1]; col[5] = Av[2];
        } else { col[0] = Av[0]; col[1] = Av[1]; col[2] = Av[2]; col[3] = 0.0; col[4] = 0.0; col[5] = 0.0; }
        add_col(col);
      }
    }
    // det of the symmetric positive semi-definite G by elimination without pivoting (det6_spd, on the packed upper triangle)
    double m[6][6];
    {
      int t = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int bb = a; bb < 6; ++bb) { m[a][bb] = G[t]; m[bb][a] = G[t]; ++t; }
    }
    double det = 1.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double piv = m[c][c];
      det *= piv;
      const double ip = (piv > 0.0) ? 1.0 / piv : 0.0;
#pragma unroll
      for (int rr = c + 1; rr < 6; ++rr) {
        const double ff = m[rr][c] * ip;
#pragma unroll
        for (int kx = c + 1; kx < 6; ++kx) m[rr][kx] = fma(-ff, m[c][kx], m[rr][kx]);
      }
    }
    f = sqrt(det > 0.0 ? det : 0.0);
  }
  S.f[lane] = on ? f : 0.0;
  WSYNC();
  // u by DoF: the sweep's central difference, the PREV value where the loop skips the DoF (HYBRID), else 0
  double u = 0.0;
  if (lane < nv && ((P.mp_prevmode >> lane) & 1u)) u = S.q[lane < 6 ? lane : lane + 1];
  if (on && side == 0) S.uo[my_i & 31] = 0.5 * (S.f[lane] - S.f[lane + 1]) / dq;   // lane 2 k holds f1, lane 2 k + 1 f2 of sweep k
  WSYNC();
  if (lane < nv) {
    bool mine = false;
#pragma unroll 1
    for (int t = 0; t < P.mp_n; ++t) mine |= (P.mp_i[t] == lane);
    if (mine) u = S.uo[lane];
  }
  if (A.u && lane < NV) A.u[(size_t)b * NV + lane] = (lane < nv) ? u : 0.0;
  if (A.q_after && lane < NQ) {
    const double a0 = S.q[lane & 31];
    A.q_after[(size_t)b * NQ + lane] = (lane < nq) ? (((P.mp_all >> lane) & 1u) ? (a0 + dq) - (dq * 2) : a0) : 0.0;
  }
}

// THREE instances per wavefront where no model of the batch has more than 21 sweeps (A1 + wx200 / px100 "MANI": 21 / 20): instance r = lane / 21,
// sweep k = lane % 21, and the lane evaluates BOTH sides of its central difference one after the other — 63 of 64 lanes busy where the kernel above
// keeps 42; the same arithmetic per evaluation (bit-identical u). Roles with more than 21 entries (configuration, sin / cos table, outputs) take
// two or three rounds of the instance's 21 lanes.
__global__ void __launch_bounds__(64) wbc_posture_par3_kernel(const PostureArgs A, const DevModel* __restrict__ models,
                                                              const WbcConfig* __restrict__ cfgs, const DevPlan* __restrict__ plans) {
  __shared__ MPSmem S3[3];
  const int lane0 = threadIdx.x;
  const bool grp = lane0 < 63;                             // (lane 63: no instance; it shadows instance 2's last lane and stores nothing)
  const int r = grp ? lane0 / 21 : 2, lane = grp ? lane0 - 21 * r : 20;
  MPSmem& S = S3[r];
  const int b_raw = 3 * (int)blockIdx.x + r;
  const bool valid = grp && b_raw < A.B;
  const int b = b_raw < A.B ? b_raw : A.B - 1;
  int mid = 0;
  if (A.model_id) { mid = A.model_id[b]; mid = mid < 0 ? 0 : (mid >= A.n_models ? A.n_models - 1 : mid); }
  const DevModel& M = models[mid];
  const DevPlan& P = plans[mid];
  const int nv = M.nv, nq = M.nq, nj = M.njoints;
  const double dq = 0.0002;
  if (grp) {
    S.q[lane] = (lane < nq) ? A.q[(size_t)b * NQ + lane] : 0.0;
    if (lane + 21 < 32) S.q[lane + 21] = (lane + 21 < nq) ? A.q[(size_t)b * NQ + lane + 21] : 0.0;
  }
  // this lane's sweep
  const int k = lane;
  const bool on = grp && k < P.mp_n;
  const int kk = on ? k : 0;
  const int my_i = P.mp_i[kk], my_qi = P.mp_qi[kk];
  const unsigned my_prev = P.mp_prev[kk];
  int chain[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) chain[c] = on ? P.mp_chain[kk][c] : -1;
  WSYNC();
  // sin / cos table: entry t < 3 (nj - 2): joint 2 + t / 3 in state t % 3; three rounds of 21 lanes
#pragma unroll 1
  for (int i = 0; i < 3; ++i) {
    const int tt = lane + 21 * i;
    const int j = 2 + tt / 3, st = tt - 3 * (tt / 3);
    if (grp && j < nj) {
      const int jt = M.jtype[j];
      if (jt >= WBC_JT_RX && jt <= WBC_JT_RZ) {
        const double a0 = S.q[M.idx_q[j]];
        const double a = (st == 0) ? a0 : ((st == 1) ? a0 + dq : (a0 + dq) - (dq * 2));
        const SinCos tsc = sincos_cw(a);
        S.sc[6 * j + 2 * st] = tsc.s; S.sc[6 * j + 2 * st + 1] = tsc.c;
      }
    }
  }
  WSYNC();
  double f1 = 0.0, f2 = 0.0;
#pragma unroll 1
  for (int side = 0; side < 2; ++side) {
  // state of configuration entry e for this lane: 0 = q, 1 = q + d, 2 = (q + d) - 2 d
  auto state_of = [&](const int e) -> int { return (e == my_qi) ? (side ? 2 : 1) : (((my_prev >> e) & 1u) ? 2 : 0); };
  auto value_of = [&](const int e) -> double {
    const double a0 = S.q[e];
    const int st = state_of(e);
    return (st == 0) ? a0 : ((st == 1) ? a0 + dq : (a0 + dq) - (dq * 2));
  };
  double f = 0.0;
  {
    // the free-flyer: R from the (possibly perturbed, not renormalised) quaternion exactly as the FK does, p = xyz
    const double qq[4] = {value_of(3), value_of(4), value_of(5), value_of(6)};
    double R1[9];
    quat_to_R(qq, R1);                                  // row-major
    double G[21];
#pragma unroll
    for (int i = 0; i < 21; ++i) G[i] = 0.0;
    auto add_col = [&](const double* c) {               // G += c c' (upper triangle, row-major packed)
      int t = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int bb = a; bb < 6; ++bb) { G[t] = fma(c[a], c[bb], G[t]); ++t; }
    };
    // ONE walk down the chain with the columns expressed at the BASE origin ((p - p_base) x axis, axis): the reference's LOCAL_WORLD_ALIGNED
    // Jacobian at the joint's origin pJ is X J_base with X = [I, -[pJ - p_base]x; 0, I], det X = 1, so det(J J') — all that f is — does not
    // depend on where the columns are expressed, and the joint's origin need not be known before the columns are formed (the first version
    // walked the chain twice for it). Origins relative to the base: the base position drops out of the arithmetic altogether, so its three
    // sweeps give f1 == f2 bit for bit and u = 0 exactly, as the reference's (and the sequential kernel's) LOCAL_WORLD_ALIGNED form does.
    {
      double X[3] = {R1[0], R1[3], R1[6]}, Y[3] = {R1[1], R1[4], R1[7]}, Z[3] = {R1[2], R1[5], R1[8]};   // columns of the parent's rotation
      double p[3] = {0.0, 0.0, 0.0};                    // origins relative to the base
      // free-flyer columns at its own origin: linear DoF (R e_i, 0), angular DoF (0, R e_i)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double* ax = (i == 0) ? X : ((i == 1) ? Y : Z);
        const double cl[6] = {ax[0], ax[1], ax[2], 0.0, 0.0, 0.0};
        add_col(cl);
        const double ca[6] = {0.0, 0.0, 0.0, ax[0], ax[1], ax[2]};
        add_col(ca);
      }
#pragma unroll 1
      for (int c = 0; c < 8; ++c) {
        const int j = chain[c];
        if (j < 0) continue;
        const int a = M.ax0[j], jt = M.jtype[j];
        const bool rev = jt >= WBC_JT_RX && jt <= WBC_JT_RZ;
        const double t0 = M.tp[j][0], t1 = M.tp[j][1], t2 = M.tp[j][2];
        const int qe = M.idx_q[j];
        const int st = state_of(qe);
        const double sn = rev ? S.sc[6 * j + 2 * st] : 0.0, cs = rev ? S.sc[6 * j + 2 * st + 1] : 1.0;
        const double pris = rev ? 0.0 : value_of(qe);
        double Av[3], Bv[3], Cv[3];                     // the axis column of the parent's rotation and its cyclic successors
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          Av[rr] = (a == 0) ? X[rr] : ((a == 1) ? Y[rr] : Z[rr]);
          Bv[rr] = (a == 0) ? Y[rr] : ((a == 1) ? Z[rr] : X[rr]);
          Cv[rr] = (a == 0) ? Z[rr] : ((a == 1) ? X[rr] : Y[rr]);
        }
        double nB[3], nC[3];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          nB[rr] = cs * Bv[rr] + sn * Cv[rr];
          nC[rr] = cs * Cv[rr] - sn * Bv[rr];
          p[rr] = p[rr] + Av[rr] * (t0 + pris) + Bv[rr] * t1 + Cv[rr] * t2;
        }
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          const double na = Av[rr], nb = nB[rr], nc = nC[rr];
          X[rr] = (a == 0) ? na : ((a == 1) ? nc : nb);
          Y[rr] = (a == 0) ? nb : ((a == 1) ? na : nc);
          Z[rr] = (a == 0) ? nc : ((a == 1) ? nb : na);
        }
        double col[6];                                  // this joint's column: revolute (p x axis, axis), prismatic (axis, 0)
        if (rev) {
          double cr[3];
          cross3(p, Av, cr);
          col[0] = cr[0]; col[1] = cr[1]; col[2] = cr[2]; col[3] = Av[0]; col[4] = Av[1]; col[5] = Av[2];
        } else { col[0] = Av[0]; col[1] = Av[1]; col[2] = Av[2]; col[3] = 0.0; col[4] = 0.0; col[5] = 0.0; }
        add_col(col);
      }
    }
    // det of the symmetric positive semi-definite G by elimination without pivoting (det6_spd, on the packed upper triangle)
    double m[6][6];
    {
      int t = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int bb = a; bb < 6; ++bb) { m[a][bb] = G[t]; m[bb][a] = G[t]; ++t; }
    }
    double det = 1.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double piv = m[c][c];
      det *= piv;
      const double ip = (piv > 0.0) ? 1.0 / piv : 0.0;
#pragma unroll
      for (int rr = c + 1; rr < 6; ++rr) {
        const double ff = m[rr][c] * ip;
#pragma unroll
        for (int kx = c + 1; kx < 6; ++kx) m[rr][kx] = fma(-ff, m[c][kx], m[rr][kx]);
      }
    }
    f = sqrt(det > 0.0 ? det : 0.0);
  }
  if (side == 0) f1 = f; else f2 = f;
  }
  // u by DoF: the sweep's central difference, the PREV value where the loop skips the DoF (HYBRID), else 0
  if (grp) { S.uo[lane] = 0.0; if (lane + 21 < 32) S.uo[lane + 21] = 0.0; S.f[lane] = 0.0; S.f[lane + 21] = 0.0; }
  WSYNC();
  if (on) { S.uo[my_i & 31] = 0.5 * (f1 - f2) / dq; S.f[my_i & 31] = 1.0; }      // (f [32]: flags "DoF swept" here)
  WSYNC();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int d = lane + 21 * i;
    if (!grp || d >= NQ) continue;
    if (d < NV) {
      double u = 0.0;
      if (d < nv && ((P.mp_prevmode >> d) & 1u)) u = S.q[d < 6 ? d : d + 1];
      if (d < nv && S.f[d & 31] != 0.0) u = S.uo[d & 31];
      if (A.u && valid) A.u[(size_t)b * NV + d] = (d < nv) ? u : 0.0;
    }
    if (A.q_after && valid) {
      const double a0 = S.q[d & 31];
      A.q_after[(size_t)b * NQ + d] = (d < nq) ? (((P.mp_all >> d) & 1u) ? (a0 + dq) - (dq * 2) : a0) : 0.0;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The tail of runWBC: updateState(joint_config, base_config, running=True) (Robot_Wrapper4.py:1397-1399, 387-428) with
// trunkWorldPos (:1297-1327). One instance per wave. In a rollout the same wave then applies the side effects qpb() has on
// the controller's reference state (:1151-1152, :995-996) and moves the targets one step along their segment.
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) USmem {
  double oMi[24 * 12];
  double q[32];
  double pf[WBC_MAX_FRAMES * 3];
  double ft[16];
};
__global__ void __launch_bounds__(64) wbc_update_kernel(const UpdateArgs A, const DevModel* __restrict__ models,
                                                        const WbcConfig* __restrict__ cfgs) {
  __shared__ USmem U;
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int mid = model_index(A.model_id, b, A.n_models);
  const DevModel& M = models[mid];
  const WbcConfig& cfg = cfgs[mid];
  // ---- every global read of the wave is issued here, before the FK (one memory round trip instead of a chain of them)
  // config = [current base xyz, base_config (IMU quaternion), new joint angles]  (:388-389)
  const int nq = M.nq;
  const bool warm = A.mode == WBC_ROLLOUT_WARMUP;   // updateState(new_config, feedback=False, running=False): the state is q_next as it is
  double c = 0.0;
  if (warm) { if (lane < nq) c = A.q_next[(size_t)b * NQ + lane]; }
  else if (lane < 3) c = A.q_cur[(size_t)b * NQ + lane];
  else if (lane < 7) c = A.imu ? A.imu[(size_t)b * 4 + (lane - 3)] : A.q_next[(size_t)b * NQ + lane];
  else if (lane < nq) c = A.q_next[(size_t)b * NQ + lane];
  const double ft = (lane < 12) ? A.foot_targets[(size_t)b * 15 + lane] : 0.0;
  const double eet = (A.ee_target && lane < 15) ? A.ee_target[(size_t)b * 15 + lane] : 0.0;
  const double ees = (A.ee_target && A.ee_step && lane < 15) ? A.ee_step[(size_t)b * 15 + lane] : 0.0;
  const double rref = (A.ee_prev_rot && A.ee_ref_rot && lane < 45) ? A.ee_ref_rot[(size_t)b * 45 + lane] : 0.0;
  const double tt = (A.trunk_target && lane < 3) ? A.trunk_target[(size_t)b * 3 + lane] : 0.0;
  const double tts = (A.trunk_target && A.trunk_step && lane < 3) ? A.trunk_step[(size_t)b * 3 + lane] : 0.0;
  const double ter = (A.trunk_prev_rot && A.trunk_ref_euler && lane < 3) ? A.trunk_ref_euler[(size_t)b * 3 + lane] : 0.0;
  int st = 0, stm = 0, it = 0, its = 0;
  if (lane == 0) {
    if (A.status_max) { st = A.status[b]; stm = A.status_max[b]; }
    if (A.iters_sum) { it = A.iters[b]; its = A.iters_sum[b]; }
  }
  const LaneConst lc = load_lane_const(M, cfg, lane);
  if (lane < 32) U.q[lane] = c;
  if (lane < 12) U.ft[lane] = ft;
  WSYNC();
  fk_levels(U.oMi, U.q, load_hdr(M), lc, lane);
  if (lane < M.nframes) {
    const double* Pj = U.oMi + lc.fj_off;
#pragma unroll
    for (int r = 0; r < 3; ++r) U.pf[3 * lane + r] = Pj[9 + r] + Pj[r] * lc.f0 + Pj[3 + r] * lc.f1 + Pj[6 + r] * lc.f2;
  }
  WSYNC();
  // trunkWorldPos: trunk_pos = WPA - WRB . BPA  (:1321-1325), evaluated uniformly
  const double* Pt = U.oMi + 12 * M.frame_joint[WBC_FR_TRUNK];   // R column-major
  double WPA[3], BPA[3], base[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double t = U.pf[3 * WBC_FR_TRUNK + i];
    WPA[i] = (U.ft[i] + U.ft[3 + i] + U.ft[6 + i] + U.ft[9 + i]) / 4;
    BPA[i] = ((U.pf[i] - t) + (U.pf[3 + i] - t) + (U.pf[6 + i] - t) + (U.pf[9 + i] - t)) / 4;
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) base[r] = WPA[r] - (Pt[r] * BPA[0] + Pt[3 + r] * BPA[1] + Pt[6 + r] * BPA[2]);
  if (warm) { base[0] = U.q[0]; base[1] = U.q[1]; base[2] = U.q[2]; }   // no estimator while warming up (running == False, :414)
  if (lane < NQ) A.q_new[(size_t)b * NQ + lane] = (lane == 0) ? base[0] : (lane == 1) ? base[1] : (lane == 2) ? base[2] : c;
  if (A.grip_trace && lane < 3) {
    // gripper_bar after the base correction: the whole tree translates rigidly with the base
    const double d = (lane == 0) ? base[0] - U.q[0] : (lane == 1) ? base[1] - U.q[1] : base[2] - U.q[2];
    A.grip_trace[(size_t)b * 3 + lane] = U.pf[3 * (WBC_FR_EE0 + 4) + lane] + d;
  }
  if (lane == 0) {
    if (A.status_max && st > stm) A.status_max[b] = st;
    if (A.iters_sum) A.iters_sum[b] = its + it;
  }
  // ---- side effects of qpb() on the reference state, then the targets move on
  if (A.ee_target && lane < 15) {
    const int e = lane / 3;
    const size_t i = (size_t)b * 15 + lane;
    if (cfg.task_ee[e] && A.prev_ee_target) A.prev_ee_target[i] = eet;               // prev_EE_pos[i] = target (:1151)
    if (A.ee_step) A.ee_target[i] = eet + ees;
  }
  if (A.ee_prev_rot && A.ee_ref_rot && lane < 45) {
    const int e = lane / 9;
    if (cfg.task_ee[e]) A.ee_prev_rot[(size_t)b * 45 + lane] = rref;                 // prev_EE_CoM_rot[i] = R* (:1152)
  }
  if (A.trunk_target && lane < 3) {
    const size_t i = (size_t)b * 3 + lane;
    if (cfg.task_trunk && A.prev_trunk_target) A.prev_trunk_target[i] = tt;          // prev_trunk_ref = target (:995)
    if (A.trunk_step) A.trunk_target[i] = tt + tts;
  }
  if (cfg.task_trunk && A.trunk_prev_rot && A.trunk_ref_euler) {                      // old_ref_trunk_rot_matrix = R* (:996)
    const SinCos a = sincos_cw(rdl(ter, 0)), bb = sincos_cw(rdl(ter, 1)), cc = sincos_cw(rdl(ter, 2));
    double Rs[9];
    Rs[0] = cc.c * bb.c; Rs[1] = cc.c * bb.s * a.s - cc.s * a.c; Rs[2] = cc.c * bb.s * a.c + cc.s * a.s;
    Rs[3] = cc.s * bb.c; Rs[4] = cc.s * bb.s * a.s + cc.c * a.c; Rs[5] = cc.s * bb.s * a.c - cc.c * a.s;
    Rs[6] = -bb.s;       Rs[7] = bb.c * a.s;                     Rs[8] = bb.c * a.c;
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 9; ++i) A.trunk_prev_rot[(size_t)b * 9 + i] = Rs[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// wbc_update_kernel for four instances per wavefront (same lane layout and FK records as the packed tick kernel): the one-instance
// kernel spends a whole wave's instruction stream on one 22-joint FK; in a roll-out that was a third of the closed-loop tick.
// Used when every plan of the batch is DevPlan.pk_update_ok (the packed FK schedule reaches every frame the estimator reads).
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) UInst {
  double oMi[24 * 12];
  double q[32];
  double sc[64];
  double pf[6 * 4];                         // feet 0..3, trunk, gripper: world positions
  double ft[16];
};
struct __attribute__((aligned(16))) USmemP { UInst I[4]; };
__global__ void __launch_bounds__(64) wbc_update_packed_kernel(const UpdateArgs A, const DevModel* __restrict__ models,
                                                               const WbcConfig* __restrict__ cfgs, const DevPlan* __restrict__ plans) {
  __shared__ USmemP UP;
  const int lane = threadIdx.x, r = lane >> 4, s = lane & 15;
  UInst& U = UP.I[r];
  const int b_raw = 4 * blockIdx.x + r;
  const bool valid = b_raw < A.B;
  const size_t b = valid ? b_raw : A.B - 1;
  int mid = 0;
  if (A.model_id) { mid = A.model_id[b]; mid = mid < 0 ? 0 : (mid >= A.n_models ? A.n_models - 1 : mid); }
  const DevModel& M = models[mid];
  const WbcConfig& cfg = cfgs[mid];
  const DevPlan& P = plans[mid];
  const int nq = M.nq;
  const bool warm = A.mode == WBC_ROLLOUT_WARMUP;
  // ---- every global read first. config = [current base xyz, base_config (IMU quaternion), new joint angles]  (:388-389)
  double c0, c1 = 0.0;
  if (warm) c0 = A.q_next[b * NQ + s];
  else if (s < 3) c0 = A.q_cur[b * NQ + s];
  else if (s < 7) c0 = A.imu ? A.imu[b * 4 + (s - 3)] : A.q_next[b * NQ + s];
  else c0 = A.q_next[b * NQ + s];
  if (16 + s < nq) c1 = A.q_next[b * NQ + 16 + s];
  const double ft = (s < 12) ? A.foot_targets[b * 15 + s] : 0.0;
  const double eet = (A.ee_target && s < 15) ? A.ee_target[b * 15 + s] : 0.0;
  const double ees = (A.ee_target && A.ee_step && s < 15) ? A.ee_step[b * 15 + s] : 0.0;
  double rref[3] = {0.0, 0.0, 0.0};
  if (A.ee_prev_rot && A.ee_ref_rot) {
#pragma unroll
    for (int h = 0; h < 3; ++h) if (16 * h + s < 45) rref[h] = A.ee_ref_rot[b * 45 + 16 * h + s];
  }
  const double tt = (A.trunk_target && s < 3) ? A.trunk_target[b * 3 + s] : 0.0;
  const double tts = (A.trunk_target && A.trunk_step && s < 3) ? A.trunk_step[b * 3 + s] : 0.0;
  const double ter = (A.trunk_prev_rot && A.trunk_ref_euler && s < 3) ? A.trunk_ref_euler[b * 3 + s] : 0.0;
  int st = 0, stm = 0, it = 0, its = 0;
  if (s == 0) {
    if (A.status_max) { st = A.status[b]; stm = A.status_max[b]; }
    if (A.iters_sum) { it = A.iters[b]; its = A.iters_sum[b]; }
  }
  DevPlan::PkJoint fkn = P.pk_fk[0][s];
  const int scq0 = P.pk_scq[(2 + s) & 31], scq1 = P.pk_scq[(18 + s) & 31];
  // frame of this lane: feet 0..3, trunk, gripper
  const int fr = (s < 4) ? WBC_FR_EE0 + s : ((s == 4) ? WBC_FR_TRUNK : WBC_FR_EE0 + 4);
  const int fjoint = M.frame_joint[fr];
  const double f0 = M.frame_p[fr][0], f1 = M.frame_p[fr][1], f2 = M.frame_p[fr][2];
  const int tjoint = M.frame_joint[WBC_FR_TRUNK];
  U.q[s] = c0; U.q[16 + s] = c1;
  if (s < 12) U.ft[s] = ft;
  WSYNC();
  const double* const qv = U.q;
  double* const oMi = U.oMi;
  if (scq0 >= 0) { const SinCos t = sincos_cw(qv[scq0]); U.sc[2 * (2 + s)] = t.s; U.sc[2 * (2 + s) + 1] = t.c; }
  if (scq1 >= 0) { const SinCos t = sincos_cw(qv[scq1]); U.sc[2 * (18 + s)] = t.s; U.sc[2 * (18 + s) + 1] = t.c; }
  if (s == 0) {
    double Rt[9];
    quat_to_R(qv + 3, Rt);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) oMi[12 + 3 * c + rr] = Rt[3 * rr + c];
    oMi[12 + 9] = qv[0]; oMi[12 + 10] = qv[1]; oMi[12 + 11] = qv[2];
  }
  WSYNC();
#pragma unroll 1
  for (int L = 0; L < 5; ++L) {             // the packed tick kernel's FK, record for record
    const DevPlan::PkJoint fk = fkn;
    if (L + 1 < 5) fkn = P.pk_fk[L + 1][s];
    const int j = fk.joint;
    if (j >= 0) {
      const bool rev = fk.rev != 0;
      const int a0 = fk.a0, a1 = fk.a1, a2 = fk.a2;
      const double* Pp = oMi + 12 * fk.parent;
      const double sn = rev ? U.sc[2 * j] : 0.0, cs = rev ? U.sc[2 * j + 1] : 1.0;
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
  if (s < 6) {
    const double* Pj = oMi + 12 * fjoint;
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) U.pf[4 * s + rr] = Pj[9 + rr] + Pj[rr] * f0 + Pj[3 + rr] * f1 + Pj[6 + rr] * f2;
  }
  WSYNC();
  // trunkWorldPos: trunk_pos = WPA - WRB . BPA  (:1321-1325), evaluated by every lane of the instance
  const double* Pt = oMi + 12 * tjoint;     // R column-major
  double WPA[3], BPA[3], base[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double t = U.pf[4 * 4 + i];
    WPA[i] = (U.ft[i] + U.ft[3 + i] + U.ft[6 + i] + U.ft[9 + i]) / 4;
    BPA[i] = ((U.pf[i] - t) + (U.pf[4 + i] - t) + (U.pf[8 + i] - t) + (U.pf[12 + i] - t)) / 4;
  }
#pragma unroll
  for (int rr = 0; rr < 3; ++rr) base[rr] = WPA[rr] - (Pt[rr] * BPA[0] + Pt[3 + rr] * BPA[1] + Pt[6 + rr] * BPA[2]);
  if (warm) { base[0] = qv[0]; base[1] = qv[1]; base[2] = qv[2]; }   // no estimator while warming up (running == False, :414)
  if (!valid) return;
  {
    double* qo = A.q_new + b * NQ;
    qo[s] = (s == 0) ? base[0] : (s == 1) ? base[1] : (s == 2) ? base[2] : c0;
    if (16 + s < NQ) qo[16 + s] = (16 + s < nq) ? c1 : 0.0;
  }
  if (A.grip_trace && s < 3) {
    const double d = (s == 0) ? base[0] - qv[0] : (s == 1) ? base[1] - qv[1] : base[2] - qv[2];
    A.grip_trace[b * 3 + s] = U.pf[4 * 5 + s] + d;
  }
  if (s == 0) {
    if (A.status_max && st > stm) A.status_max[b] = st;
    if (A.iters_sum) A.iters_sum[b] = its + it;
  }
  // ---- side effects of qpb() on the reference state, then the targets move on (as wbc_update_kernel)
  if (A.ee_target && s < 15) {
    const int e = s / 3;
    if (cfg.task_ee[e] && A.prev_ee_target) A.prev_ee_target[b * 15 + s] = eet;       // prev_EE_pos[i] = target (:1151)
    if (A.ee_step) A.ee_target[b * 15 + s] = eet + ees;
  }
  if (A.ee_prev_rot && A.ee_ref_rot) {
#pragma unroll
    for (int h = 0; h < 3; ++h) {
      const int i = 16 * h + s;
      if (i < 45 && cfg.task_ee[i / 9]) A.ee_prev_rot[b * 45 + i] = rref[h];          // prev_EE_CoM_rot[i] = R* (:1152)
    }
  }
  if (A.trunk_target && s < 3) {
    if (cfg.task_trunk && A.prev_trunk_target) A.prev_trunk_target[b * 3 + s] = tt;   // prev_trunk_ref = target (:995)
    if (A.trunk_step) A.trunk_target[b * 3 + s] = tt + tts;
  }
  if (A.trunk_prev_rot && A.trunk_ref_euler && __ballot(cfg.task_trunk != 0)) {      // old_ref_trunk_rot_matrix = R* (:996), with the trunk task on
    const SinCos t = sincos_cw(ter);          // lanes 0..2 of the row: roll, pitch, yaw of the reference
    const int rb = lane & 48;
    const double sa = bperm(t.s, rb), ca = bperm(t.c, rb), sb = bperm(t.s, rb + 1), cb = bperm(t.c, rb + 1), sc_ = bperm(t.s, rb + 2), cc = bperm(t.c, rb + 2);
    double Rs[9];
    Rs[0] = cc * cb; Rs[1] = cc * sb * sa - sc_ * ca; Rs[2] = cc * sb * ca + sc_ * sa;
    Rs[3] = sc_ * cb; Rs[4] = sc_ * sb * sa + cc * ca; Rs[5] = sc_ * sb * ca - cc * sa;
    Rs[6] = -sb;      Rs[7] = cb * sa;                 Rs[8] = cb * ca;
    if (valid && cfg.task_trunk && s < 9) {
      double v = Rs[0];
#pragma unroll
      for (int i = 1; i < 9; ++i) v = (s == i) ? Rs[i] : v;
      A.trunk_prev_rot[b * 9 + s] = v;
    }
  }
}



int launch_integrate(const IntegrateArgs& a, int grid, void* stream) {
  hipLaunchKernelGGL(wbc_integrate_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, a);
  return check_launch("integrate");
}

int launch_posture(const PostureArgs& a, int grid, void* stream) {
  hipLaunchKernelGGL(wbc_posture_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs);
  return check_launch("posture");
}

int launch_posture_par(const PostureArgs& a, int grid, void* stream, int three) {
  if (three) hipLaunchKernelGGL(wbc_posture_par3_kernel, dim3((a.B + 2) / 3), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  else hipLaunchKernelGGL(wbc_posture_par_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  return check_launch("posture_par");
}

int launch_update(const UpdateArgs& a, int grid, void* stream) {
  hipLaunchKernelGGL(wbc_update_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs);
  return check_launch("update");
}

int launch_update_packed(const UpdateArgs& a, void* stream) {
  hipLaunchKernelGGL(wbc_update_packed_kernel, dim3((a.B + 3) / 4), dim3(64), 0, (hipStream_t)stream, a, a.models, a.cfgs, a.plans);
  return check_launch("update_packed");
}
#endif

}  // namespace wbc
