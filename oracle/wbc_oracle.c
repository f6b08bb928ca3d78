/*
 * wbc_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C fp64 restatement of the reference's per-tick hot path, used ONLY by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg to check / time against the HIP path.
 * Nothing under mech5845m-wbc-for-legged-manipulator_amd/ may import, link or call this file.
 *
 * PARITY PINNING: the arithmetic of the reference lives in pinocchio and qpOASES,
 * neither vendored, pinned nor installable here, and the reference has no test suite.
 *   - FK / Jacobians: weakly pinned by the reference's only numeric dump, tests_NOT_FOR_USE/Jacobians.py
 *     (committed as tests/golden/jacobians_kat.json), plus finite-difference and identity certificates.
 *   - quaternion / Euler helpers: pinned against scipy.spatial.transform.Rotation (importable here).
 *   - task/constraint assembly and QP solutions: PARITY UNPINNED (no expected values exist anywhere in
 *     the reference). The QP restatement is certified by KKT residuals and scipy.optimize cross-solves;
 *     H > 0 makes the minimiser unique, hence solver independent. Since round 4 the solve ends with one step of iterative
 *     refinement (wrappers/QP_Wrapper.py:37 numRefinementSteps; qp_refine below) whose residual comes from the least-squares
 *     data A, b — the answer is the exact optimum of min 1/2 |A x - b|^2 on the final working set to ~1e-13 (tests compare
 *     with rational arithmetic), not the optimum of the rounded H = fl(A'A), which differs from it by ~1e-7 on the benchmark tick.
 *
 * Each function cites the reference file:line it follows (paths relative to /root/reference).
 * Third-party semantics restated: pinocchio (≈2.5–2.6, unpinned) forwardKinematics,
 * computeJointJacobians, getFrameJacobian, jacobianCenterOfMass, integrate; qpOASES is replaced by
 * the Goldfarb–Idnani dual active-set method (Math. Prog. 27, 1983 — the algorithm behind "quadprog",
 * which BASELINE.json names as the CPU reference), exact for strictly convex QPs.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../include/wbc.h"

#define NV WBC_V_STRIDE
#define NQS WBC_Q_STRIDE

/* ---------------------------------------------------------------- small linear algebra */
static void m3_mul(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
static void m3_vec(const double* A, const double* v, double* r) {
  for (int i = 0; i < 3; ++i) r[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
static void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

/* Eigen::Quaternion::toRotationMatrix (no normalisation) — what pinocchio's free-flyer calc uses. */
void orc_quat_to_R(const double* q /*xyzw*/, double* R) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* scipy Rotation.from_matrix(M).as_euler('xyz') for a proper rotation: extrinsic x-y-z,
 * M = Rz(c) Ry(b) Rx(a) -> (a, b, c).  Used at Robot_Wrapper4.py:714-715, 363-367, 382-383.
 * The pitch is written atan2(-M20, |(M21, M22)|) (= -asin(M20) on a proper rotation). On a matrix that is NOT
 * orthonormal — only the "MANI" posture mode produces one, by perturbing the free-flyer quaternion —
 * scipy first projects M to a rotation, by a method that changed between scipy releases (unit quaternion, later SVD)
 * and that the reference does not pin: that corner is "parity unpinned"; oracle and device share this formula. */
void orc_R_to_euler_xyz(const double* M, double* e) {
  e[0] = atan2(M[7], M[8]);
  e[1] = atan2(-M[6], sqrt(M[7] * M[7] + M[8] * M[8]));
  e[2] = atan2(M[3], M[0]);
}
/* scipy Rotation.from_euler('xyz', e).as_matrix() (Robot_Wrapper4.py:968-969, 1101-1102). */
void orc_euler_xyz_to_R(const double* e, double* R) {
  const double ca = cos(e[0]), sa = sin(e[0]), cb = cos(e[1]), sb = sin(e[1]), cc = cos(e[2]), sc = sin(e[2]);
  R[0] = cc * cb; R[1] = cc * sb * sa - sc * ca; R[2] = cc * sb * ca + sc * sa;
  R[3] = sc * cb; R[4] = sc * sb * sa + cc * ca; R[5] = sc * sb * ca - cc * sa;
  R[6] = -sb;     R[7] = cb * sa;                R[8] = cb * ca;
}
/* scipy Rotation.from_matrix(M).as_quat() (Robot_Wrapper4.py:964-965): largest-of-(diag, trace) branch. */
void orc_R_to_quat(const double* M, double* q) {
  const double tr = M[0] + M[4] + M[8];
  double dec[4] = {M[0], M[4], M[8], tr};
  int c = 0;
  for (int i = 1; i < 4; ++i)
    if (dec[i] > dec[c]) c = i;
  if (c != 3) {
    const int i = c, j = (i + 1) % 3, k = (j + 1) % 3;
    q[i] = 1 - tr + 2 * M[3 * i + i];
    q[j] = M[3 * j + i] + M[3 * i + j];
    q[k] = M[3 * k + i] + M[3 * i + k];
    q[3] = M[3 * k + j] - M[3 * j + k];
  } else {
    q[0] = M[7] - M[5]; q[1] = M[2] - M[6]; q[2] = M[3] - M[1]; q[3] = 1 + tr;
  }
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) q[i] /= n;
}
static void quat_mul(const double* a, const double* b, double* r) { /* xyzw, Hamilton */
  const double x1 = a[0], y1 = a[1], z1 = a[2], w1 = a[3], x2 = b[0], y2 = b[1], z2 = b[2], w2 = b[3];
  r[0] = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2;
  r[1] = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2;
  r[2] = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2;
  r[3] = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2;
}
/* scipy Rotation.from_euler('xyz', e).as_quat() (Robot_Wrapper4.py:968-970): q = qz * qy * qx. */
void orc_euler_xyz_to_quat(const double* e, double* q) {
  const double qx[4] = {sin(e[0] / 2), 0, 0, cos(e[0] / 2)};
  const double qy[4] = {0, sin(e[1] / 2), 0, cos(e[1] / 2)};
  const double qz[4] = {0, 0, sin(e[2] / 2), cos(e[2] / 2)};
  double t[4];
  quat_mul(qy, qx, t);
  quat_mul(qz, t, q);
}

/* ---------------------------------------------------------------- kinematics (pinocchio semantics) */

/* pin.forwardKinematics (Robot_Wrapper4.py:400): oMi[j] = oMi[parent] * placement_j * jointTransform(q_j).
 * oMi is [njoints][12] = R (row-major 9) then p (3). */
void orc_fk(const WbcModelBlob* m, const double* q, double* oMi) {
  static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  memcpy(oMi, I3, sizeof I3);
  oMi[9] = oMi[10] = oMi[11] = 0;
  for (int j = 1; j < m->njoints; ++j) {
    const double* Mp = oMi + 12 * m->parent[j];
    double Rl[9], pl[3] = {0, 0, 0}, Rt[9], pt[3];
    const int t = m->jtype[j];
    const double* qj = q + m->idx_q[j];
    memcpy(Rl, I3, sizeof I3);
    if (t == WBC_JT_FF) {
      orc_quat_to_R(qj + 3, Rl);
      pl[0] = qj[0]; pl[1] = qj[1]; pl[2] = qj[2];
    } else if (t >= WBC_JT_RX && t <= WBC_JT_RZ) {
      const double c = cos(qj[0]), s = sin(qj[0]);
      if (t == WBC_JT_RX) { Rl[4] = c; Rl[5] = -s; Rl[7] = s; Rl[8] = c; }
      if (t == WBC_JT_RY) { Rl[0] = c; Rl[2] = s; Rl[6] = -s; Rl[8] = c; }
      if (t == WBC_JT_RZ) { Rl[0] = c; Rl[1] = -s; Rl[3] = s; Rl[4] = c; }
    } else {
      pl[t - WBC_JT_PX] = qj[0];
    }
    /* liMi = placement * joint transform */
    double Rli[9], pli[3];
    m3_mul(m->place_R[j], Rl, Rli);
    m3_vec(m->place_R[j], pl, pli);
    for (int i = 0; i < 3; ++i) pli[i] += m->place_p[j][i];
    m3_mul(Mp, Rli, Rt);
    m3_vec(Mp, pli, pt);
    double* Mo = oMi + 12 * j;
    memcpy(Mo, Rt, sizeof Rt);
    for (int i = 0; i < 3; ++i) Mo[9 + i] = Mp[9 + i] + pt[i];
  }
}

/* pin.computeJointJacobians (Robot_Wrapper4.py:403): data.J, 6 x nv, WORLD frame; column of joint j's
 * DoF = oMi[j].act(S): rows 0-2 linear (velocity of the body point passing through the world origin),
 * rows 3-5 angular. J is [6][NV] with unused columns zero. */
void orc_joint_jacobians(const WbcModelBlob* m, const double* oMi, double* J) {
  memset(J, 0, sizeof(double) * 6 * NV);
  for (int j = 1; j < m->njoints; ++j) {
    const double* R = oMi + 12 * j;
    const double* p = R + 9;
    const int t = m->jtype[j], v = m->idx_v[j];
    if (t == WBC_JT_FF) {
      for (int i = 0; i < 3; ++i) {
        const double a[3] = {R[i], R[3 + i], R[6 + i]}; /* R e_i */
        double pxa[3];
        cross3(p, a, pxa);
        for (int r = 0; r < 3; ++r) {
          J[r * NV + v + i] = a[r];
          J[r * NV + v + 3 + i] = pxa[r];
          J[(3 + r) * NV + v + 3 + i] = a[r];
        }
      }
    } else if (t >= WBC_JT_RX && t <= WBC_JT_RZ) {
      const int k = t - WBC_JT_RX;
      const double a[3] = {R[k], R[3 + k], R[6 + k]};
      double pxa[3];
      cross3(p, a, pxa);
      for (int r = 0; r < 3; ++r) { J[r * NV + v] = pxa[r]; J[(3 + r) * NV + v] = a[r]; }
    } else {
      const int k = t - WBC_JT_PX;
      for (int r = 0; r < 3; ++r) J[r * NV + v] = R[3 * r + k];
    }
  }
}

/* is joint `anc` on the path root..j (inclusive)? */
static int supports(const WbcModelBlob* m, int anc, int j) {
  while (j > 0) {
    if (j == anc) return 1;
    j = m->parent[j];
  }
  return 0;
}

/* pin.updateFramePlacements (Robot_Wrapper4.py:405): oMf = oMi[parent] * placement. */
void orc_frame_placement(const WbcModelBlob* m, const double* oMi, int f, double* Mf) {
  const double* Mj = oMi + 12 * m->frame_joint[f];
  double pt[3];
  m3_mul(Mj, m->frame_R[f], Mf);
  m3_vec(Mj, m->frame_p[f], pt);
  for (int i = 0; i < 3; ++i) Mf[9 + i] = Mj[9 + i] + pt[i];
}

/* pin.getFrameJacobian(model, data, frame, rf) (Robot_Wrapper4.py:480, 488, 709, 758), and with
 * joint >= 0 pin.getJointJacobian (tests_NOT_FOR_USE/Jacobians.py dumps). rf: 0 WORLD, 1 LOCAL,
 * 2 LOCAL_WORLD_ALIGNED. Jf is [6][NV]. */
void orc_frame_jacobian(const WbcModelBlob* m, const double* oMi, const double* J, int frame, int joint, int rf,
                        double* Jf) {
  double Mf[12];
  int jf;
  if (joint >= 0) { jf = joint; memcpy(Mf, oMi + 12 * joint, sizeof Mf); }
  else { jf = m->frame_joint[frame]; orc_frame_placement(m, oMi, frame, Mf); }
  memset(Jf, 0, sizeof(double) * 6 * NV);
  for (int j = 1; j < m->njoints; ++j) {
    if (!supports(m, j, jf)) continue;
    const int nvj = (m->jtype[j] == WBC_JT_FF) ? 6 : 1;
    for (int c = m->idx_v[j]; c < m->idx_v[j] + nvj; ++c) {
      double lin[3] = {J[c], J[NV + c], J[2 * NV + c]}, ang[3] = {J[3 * NV + c], J[4 * NV + c], J[5 * NV + c]};
      if (rf == 2 || rf == 1) { /* shift the reference point to the frame origin: lin - p x ang */
        double pxw[3];
        cross3(Mf + 9, ang, pxw);
        for (int r = 0; r < 3; ++r) lin[r] -= pxw[r];
      }
      if (rf == 1) { /* rotate into the frame: R^T */
        double l2[3], a2[3];
        for (int r = 0; r < 3; ++r) {
          l2[r] = Mf[r] * lin[0] + Mf[3 + r] * lin[1] + Mf[6 + r] * lin[2];
          a2[r] = Mf[r] * ang[0] + Mf[3 + r] * ang[1] + Mf[6 + r] * ang[2];
        }
        memcpy(lin, l2, sizeof l2); memcpy(ang, a2, sizeof a2);
      }
      for (int r = 0; r < 3; ++r) { Jf[r * NV + c] = lin[r]; Jf[(3 + r) * NV + c] = ang[r]; }
    }
  }
}

/* pin.jacobianCenterOfMass (Robot_Wrapper4.py:670, Robot_Wrapper2.py:601): data.com[0] and the 3 x nv
 * CoM Jacobian: column k of joint j = (m_subtree(j)/M) * (lin_k + ang_k x c_subtree(j)). */
void orc_com(const WbcModelBlob* m, const double* oMi, const double* J, double* com, double* Jcom) {
  double ms[WBC_MAX_JOINTS], mc[WBC_MAX_JOINTS][3];
  for (int j = 0; j < m->njoints; ++j) {
    double c[3];
    m3_vec(oMi + 12 * j, m->com[j], c);
    ms[j] = m->mass[j];
    for (int i = 0; i < 3; ++i) mc[j][i] = m->mass[j] * (c[i] + oMi[12 * j + 9 + i]);
  }
  for (int j = m->njoints - 1; j >= 1; --j) { /* parents have smaller indices */
    const int p = m->parent[j];
    ms[p] += ms[j];
    for (int i = 0; i < 3; ++i) mc[p][i] += mc[j][i];
  }
  const double M = ms[0];
  for (int i = 0; i < 3; ++i) com[i] = mc[0][i] / M;
  if (!Jcom) return;
  memset(Jcom, 0, sizeof(double) * 3 * NV);
  for (int j = 1; j < m->njoints; ++j) {
    const int nvj = (m->jtype[j] == WBC_JT_FF) ? 6 : 1;
    double cs[3] = {0, 0, 0};
    if (ms[j] > 0) for (int i = 0; i < 3; ++i) cs[i] = mc[j][i] / ms[j];
    for (int c = m->idx_v[j]; c < m->idx_v[j] + nvj; ++c) {
      const double lin[3] = {J[c], J[NV + c], J[2 * NV + c]}, ang[3] = {J[3 * NV + c], J[4 * NV + c], J[5 * NV + c]};
      double wxc[3];
      cross3(ang, cs, wxc);
      for (int r = 0; r < 3; ++r) Jcom[r * NV + c] = (ms[j] / M) * (lin[r] + wxc[r]);
    }
  }
}

/* pin.integrate(model, q, v) with v = qdot*dt (Robot_Wrapper4.py:441). Free-flyer: M+ = M exp6(v)
 * (body-frame twist), quaternion from the rotation matrix, sign kept continuous, first-order
 * renormalised (pinocchio SpecialEuclideanOperationTpl<3>::integrate_impl); 1-DoF joints: q + v. */
void orc_integrate(const WbcModelBlob* m, const double* q, const double* v, double* qn) {
  for (int j = 1; j < m->njoints; ++j) {
    const int iq = m->idx_q[j], iv = m->idx_v[j];
    if (m->jtype[j] != WBC_JT_FF) { qn[iq] = q[iq] + v[iv]; continue; }
    const double* vl = v + iv; const double* w = v + iv + 3;
    double R0[9], Re[9], pe[3], R1[9], pr[3];
    orc_quat_to_R(q + iq + 3, R0);
    const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], t = sqrt(t2);
    double a, b, c; /* a = sin t / t, b = (1 - cos t)/t^2, c = (1 - a)/t^2 */
    if (t < 1e-4) { a = 1 - t2 / 6; b = 0.5 - t2 / 24; c = 1.0 / 6 - t2 / 120; }
    else { a = sin(t) / t; b = (1 - cos(t)) / t2; c = (1 - a) / t2; }
    /* exp3: I + a [w]x + b [w]x^2 */
    const double wx = w[0], wy = w[1], wz = w[2];
    Re[0] = 1 - b * (wy * wy + wz * wz); Re[1] = -a * wz + b * wx * wy;       Re[2] = a * wy + b * wx * wz;
    Re[3] = a * wz + b * wx * wy;        Re[4] = 1 - b * (wx * wx + wz * wz); Re[5] = -a * wx + b * wy * wz;
    Re[6] = -a * wy + b * wx * wz;       Re[7] = a * wx + b * wy * wz;        Re[8] = 1 - b * (wx * wx + wy * wy);
    double wxv[3];
    cross3(w, vl, wxv);
    const double wv = w[0] * vl[0] + w[1] * vl[1] + w[2] * vl[2];
    for (int i = 0; i < 3; ++i) pe[i] = a * vl[i] + b * wxv[i] + c * wv * w[i];
    m3_mul(R0, Re, R1);
    m3_vec(R0, pe, pr);
    for (int i = 0; i < 3; ++i) qn[iq + i] = q[iq + i] + pr[i];
    /* Eigen quaternion-from-matrix (trace branch first), then sign + firstOrderNormalize */
    double qq[4];
    const double tr = R1[0] + R1[4] + R1[8];
    if (tr > 0) {
      double s = sqrt(tr + 1.0);
      qq[3] = 0.5 * s; s = 0.5 / s;
      qq[0] = (R1[7] - R1[5]) * s; qq[1] = (R1[2] - R1[6]) * s; qq[2] = (R1[3] - R1[1]) * s;
    } else {
      int i = 0;
      if (R1[4] > R1[0]) i = 1;
      if (R1[8] > R1[3 * i + i]) i = 2;
      const int jj = (i + 1) % 3, k = (jj + 1) % 3;
      double s = sqrt(R1[3 * i + i] - R1[3 * jj + jj] - R1[3 * k + k] + 1.0);
      qq[i] = 0.5 * s; s = 0.5 / s;
      qq[3] = (R1[3 * k + jj] - R1[3 * jj + k]) * s;
      qq[jj] = (R1[3 * jj + i] + R1[3 * i + jj]) * s;
      qq[k] = (R1[3 * k + i] + R1[3 * i + k]) * s;
    }
    const double* q0 = q + iq + 3;
    if (qq[0] * q0[0] + qq[1] * q0[1] + qq[2] * q0[2] + qq[3] * q0[3] < 0) for (int i = 0; i < 4; ++i) qq[i] = -qq[i];
    const double n2 = qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3];
    const double f = (3 - n2) / 2;
    for (int i = 0; i < 4; ++i) qn[iq + 3 + i] = qq[i] * f;
  }
}

/* ---------------------------------------------------------------- assembly */

/* row counts under the current switches: qpA() (Robot_Wrapper4.py:839-876, 1271-1280) and
 * findConstraints() (Robot_Wrapper4.py:764-836). */
int orc_task_rows(const WbcConfig* c) {
  int m = 0;
  for (int i = 0; i < WBC_NEE; ++i) m += c->task_ee[i] ? 6 : 0;
  m += c->task_trunk ? 6 : 0;
  m += c->task_com ? 3 : 0;
  m += c->task_joint ? NV : 0;
  return m;
}
int orc_constraint_rows(const WbcConfig* c) {
  int p = 0;
  p += c->con_com ? 2 : 0;
  p += c->con_trunk ? 4 : 0;
  for (int i = 0; i < WBC_NEE; ++i) p += c->con_ee[i] ? 3 : 0;
  return p;
}


/* One instance of the task stack and constraints. Outputs (any may be NULL):
 * A [m][NV], bv [m], C [p][NV], Clb/Cub [p], lb/ub [NV], H [NV][NV], g [NV].
 * Columns >= model nv are padded: A = 0, H_dd = 1, bounds 0. */
void orc_posture_target(const WbcModelBlob* m, const double* q_in, int mode, int arm_base_id, int literal,
                        double* u, double* q_after);
/* q_con_used (27, optional): the configuration the constraints, bounds and the integration of this tick see. */
void orc_assemble_one(const WbcModelBlob* m, const WbcConfig* c, const WbcTickIn* in, int b, double dt,
                      double* A, double* bv, double* C, double* Clb, double* Cub, double* lb, double* ub,
                      double* H, double* g, double* q_con_used) {
  const double* q = in->q + (size_t)b * NQS;
  const double* qc = in->q_con ? in->q_con + (size_t)b * NQS : q;
  double upost[NV], qleft[NQS];
  memset(upost, 0, sizeof upost);
  if (c->task_joint >= WBC_JOINT_MANI) {
    if (in->posture_u) memcpy(upost, in->posture_u + (size_t)b * NV, sizeof upost);
    else if (c->task_joint != WBC_JOINT_CUSTOM) {   /* qpJointb MANI / HYBRID (Robot_Wrapper4.py:1220-1260) */
      orc_posture_target(m, q, c->task_joint, c->arm_base_id, c->posture_literal, upost, qleft);
      if (c->posture_literal && !in->q_con) qc = qleft;   /* the perturbed state leaks into the rest of the tick (C.4) */
    }
  }
  if (q_con_used) memcpy(q_con_used, qc, sizeof(double) * NQS);
  double oMi[WBC_MAX_JOINTS * 12], J[6 * NV], Jf[6 * NV];
  const int mrows = orc_task_rows(c), prows = orc_constraint_rows(c), nv = m->nv;
  /* task stack on the stack (at most 5 x 6 + 6 + 3 + 26 = 65 rows <= WBC_MAX_M): two heap allocations per instance were
   * a measurable part of a tick and serialised the OpenMP threads in the allocator (the cpu_baseline leg of bench.py) */
  double At[WBC_MAX_M * NV], bt[WBC_MAX_M];
  memset(At, 0, sizeof(double) * (size_t)(mrows ? mrows : 1) * NV);
  memset(bt, 0, sizeof(double) * (size_t)(mrows ? mrows : 1));
  int row = 0;

  orc_fk(m, q, oMi);                 /* updateState: Robot_Wrapper4.py:400-405 */
  orc_joint_jacobians(m, oMi, J);
  double Mtrunk[12];
  orc_frame_placement(m, oMi, WBC_FR_TRUNK, Mtrunk);

  /* --- Cartesian EE tasks in order FR, FL, RR, RL, GRIP: qpCartesianA/B (Robot_Wrapper4.py:845-863, 1165-1183) */
  for (int e = 0; e < WBC_NEE; ++e) {
    if (!c->task_ee[e]) continue;
    double Mf[12];
    orc_frame_placement(m, oMi, WBC_FR_EE0 + e, Mf);
    /* endEffectorA2 (Robot_Wrapper4.py:474-484): A = EE_weight[i] . (J_LWA * cart_task_weight) */
    orc_frame_jacobian(m, oMi, J, WBC_FR_EE0 + e, -1, 2, Jf);
    for (int r = 0; r < 6; ++r)
      for (int k = 0; k < NV; ++k) At[(row + r) * NV + k] = c->ee_W[e][r] * (Jf[r * NV + k] * c->ee_w[e]);
    /* calcTargetVelEE3 (Robot_Wrapper4.py:1052-1157) */
    const double* xt = in->ee_target + ((size_t)b * WBC_NEE + e) * 3;
    const double* xp = in->prev_ee_target + ((size_t)b * WBC_NEE + e) * 3;
    double vel[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; ++i) {
      const double ref_vel = (xt[i] - xp[i]) / dt;                                     /* :1063 */
      vel[i] = ref_vel + c->ee_gain[e][i] * ((xt[i] - Mf[9 + i]) / dt);                /* :1070 */
    }
    if (in->ee_ref_rot && in->ee_prev_rot) {
      /* skew = ((R* - R*_prev)/dt) R*^T, omega = vee(skew) (:1125-1128); the quaternion feedback term
       * computed at :1108-1118 is overwritten at :1133 and never reaches b. */
      const double* Rs = in->ee_ref_rot + ((size_t)b * WBC_NEE + e) * 9;
      const double* Rp = in->ee_prev_rot + ((size_t)b * WBC_NEE + e) * 9;
      double D[9], S[9], RsT[9];
      for (int i = 0; i < 9; ++i) D[i] = (Rs[i] - Rp[i]) / dt;
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) RsT[3 * i + j] = Rs[3 * j + i];
      m3_mul(D, RsT, S);
      vel[3] = S[7]; vel[4] = S[2]; vel[5] = S[3];
    }
    for (int r = 0; r < 6; ++r) bt[row + r] = vel[r] * c->ee_w[e];                      /* EndEffectorB2 :907-910 */
    row += 6;
  }
  /* --- trunk task: trunkA (Robot_Wrapper4.py:487-490, WORLD frame) and calcTargetVelTrunk2 (:948-1015) */
  if (c->task_trunk) {
    orc_frame_jacobian(m, oMi, J, WBC_FR_TRUNK, -1, 0, Jf);
    for (int r = 0; r < 6; ++r)
      for (int k = 0; k < NV; ++k) At[(row + r) * NV + k] = (c->trunk_W[r] * Jf[r * NV + k]) * c->trunk_w;
    const double* xt = in->trunk_target + (size_t)b * 3;
    const double* xp = in->prev_trunk_target + (size_t)b * 3;
    double vel[6];
    for (int i = 0; i < 3; ++i)
      vel[i] = (xt[i] - xp[i]) / dt + c->trunk_gain[i] * ((xt[i] - Mtrunk[9 + i]) / dt);  /* :955-958 */
    double fq[4], rq[4], Rs[9], qe[3];
    orc_R_to_quat(Mtrunk, fq);                                                           /* :964-965 */
    const double* er = in->trunk_ref_euler + (size_t)b * 3;
    orc_euler_xyz_to_R(er, Rs);
    orc_euler_xyz_to_quat(er, rq);                                                       /* :968-970 */
    qe[0] = fq[3] * rq[0] - fq[0] * rq[3] + fq[1] * rq[2] - fq[2] * rq[1];               /* :974 */
    qe[1] = fq[3] * rq[1] - fq[1] * rq[3] - fq[0] * rq[2] + fq[2] * rq[0];               /* :975 */
    qe[2] = fq[3] * rq[2] - fq[3] * rq[2] + fq[0] * rq[1] - fq[1] * rq[0];               /* :976 (sic) */
    const double* Ro = in->trunk_prev_rot + (size_t)b * 9;
    double D[9], S[9];
    for (int i = 0; i < 9; ++i) D[i] = (Rs[i] - Ro[i]) / dt;
    m3_mul(D, Rs, S);                                                                    /* :984 (R*, not R*^T) */
    vel[3] = S[7] + c->trunk_gain[3] * qe[0];
    vel[4] = S[2] + c->trunk_gain[4] * qe[1];
    vel[5] = S[3] + c->trunk_gain[5] * qe[2];
    for (int r = 0; r < 6; ++r) bt[row + r] = vel[r] * c->trunk_w;                       /* TrunkB :914-920 */
    row += 6;
  }
  /* --- CoM task of Robot_Wrapper2: comJacobian (Robot_Wrapper2.py:600-603), cartesianTargetCoM (:661-668) */
  double com[3], Jcom[3 * NV];
  if (c->task_com || c->con_com) orc_com(m, oMi, J, com, Jcom);
  if (c->task_com) {
    const double* ct = in->com_target + (size_t)b * 3;
    const double* cv = in->com_target_vel + (size_t)b * 3;
    for (int r = 0; r < 3; ++r) {
      for (int k = 0; k < NV; ++k) At[(row + r) * NV + k] = c->com_W[r] * Jcom[r * NV + k];
      bt[row + r] = cv[r] + c->com_gain[r] * (ct[r] - com[r]);
    }
    row += 3;
  }
  /* --- posture rows: qpJointA (Robot_Wrapper4.py:1199-1206), qpJointb (:1209-1268) */
  if (c->task_joint) {
    const double d = (1.0 / nv) * c->joint_w;
    for (int k = 0; k < nv; ++k) {
      At[(row + k) * NV + k] = d;
      double u = 0.0;                                               /* True: Tikhonov :1212-1213 */
      if (c->task_joint == WBC_JOINT_PREV) u = (k < 6) ? q[k] : q[k + 1]; /* np.delete(q, 6) :1216-1217 */
      if (c->task_joint >= WBC_JOINT_MANI) u = upost[k];                  /* :1220-1260 */
      bt[row + k] = (1.0 / nv) * u * c->joint_w;                    /* :1262-1266 */
    }
    row += NV;
  }

  /* --- H = A'A, g = -A'b: QP.__init__ (QP_Wrapper.py:17-18) */
  if (H) {
    for (int i = 0; i < NV; ++i)
      for (int k = 0; k < NV; ++k) {
        double s = 0;
        for (int r = 0; r < mrows; ++r) s += At[r * NV + i] * At[r * NV + k];
        H[i * NV + k] = s;
      }
    for (int k = nv; k < NV; ++k) H[k * NV + k] = 1.0;
  }
  if (g) for (int k = 0; k < NV; ++k) {
    double s = 0;
    for (int r = 0; r < mrows; ++r) s += At[r * NV + k] * bt[r];
    g[k] = -s;
  }
  if (A) memcpy(A, At, sizeof(double) * (size_t)mrows * NV);
  if (bv) memcpy(bv, bt, sizeof(double) * (size_t)mrows);

  /* --- constraints in order CoM, Trunk, FR, FL, RR, RL, Grip: findConstraints (Robot_Wrapper4.py:764-836) */
  if (qc != q) {   /* robot_data now belongs to the configuration qpJointb left behind */
    q = qc;
    orc_fk(m, q, oMi);
    orc_joint_jacobians(m, oMi, J);
    orc_frame_placement(m, oMi, WBC_FR_TRUNK, Mtrunk);
    if (c->con_com) orc_com(m, oMi, J, com, Jcom);
  }
  int prow = 0;
  double Cl[WBC_MAX_P * NV], cl[WBC_MAX_P], cu[WBC_MAX_P];
  memset(Cl, 0, sizeof Cl);
  if (c->con_com) { /* CoMConstraint (Robot_Wrapper4.py:669-694); EE_frame_pos[1] = FL, [2] = RR */
    double Mfl[12], Mrr[12];
    orc_frame_placement(m, oMi, WBC_FR_EE0 + 1, Mfl);
    orc_frame_placement(m, oMi, WBC_FR_EE0 + 2, Mrr);
    for (int r = 0; r < 2; ++r) {
      memcpy(Cl + (prow + r) * NV, Jcom + r * NV, sizeof(double) * NV);
      cl[prow + r] = ((Mrr[9 + r] - com[r]) / dt) * c->com_box_scale;
      cu[prow + r] = ((Mfl[9 + r] - com[r]) / dt) * c->com_box_scale;
    }
    prow += 2;
  }
  if (c->con_trunk) { /* trunkConstraint (Robot_Wrapper4.py:707-754): LWA rows 2..5 */
    orc_frame_jacobian(m, oMi, J, WBC_FR_TRUNK, -1, 2, Jf);
    double eul[3];
    orc_R_to_euler_xyz(Mtrunk, eul);
    const double* bc = in->trunk_box_center + (size_t)b * 4;
    const double cur[4] = {Mtrunk[11], eul[0], eul[1], eul[2]};
    const double var[4] = {bc[0] * c->trunk_box_z_frac, c->trunk_box_ang, c->trunk_box_ang, c->trunk_box_ang};
    for (int r = 0; r < 4; ++r) {
      memcpy(Cl + (prow + r) * NV, Jf + (2 + r) * NV, sizeof(double) * NV);
      cl[prow + r] = (((bc[r] - var[r]) - cur[r]) / dt) * c->trunk_box_scale;
      cu[prow + r] = (((bc[r] + var[r]) - cur[r]) / dt) * c->trunk_box_scale;
    }
    prow += 4;
  }
  for (int e = 0; e < WBC_NEE; ++e) { /* EEConstraint (Robot_Wrapper4.py:757-761): WORLD rows 0..2, 0 <= . <= 0 */
    if (!c->con_ee[e]) continue;
    orc_frame_jacobian(m, oMi, J, WBC_FR_EE0 + e, -1, 0, Jf);
    for (int r = 0; r < 3; ++r) {
      memcpy(Cl + (prow + r) * NV, Jf + r * NV, sizeof(double) * NV);
      cl[prow + r] = 0; cu[prow + r] = 0;
    }
    prow += 3;
  }
  if (C) memcpy(C, Cl, sizeof(double) * (size_t)prows * NV);
  if (Clb) memcpy(Clb, cl, sizeof(double) * (size_t)prows);
  if (Cub) memcpy(Cub, cu, sizeof(double) * (size_t)prows);

  /* --- velDamperJointConstraints (Robot_Wrapper4.py:572-637). The index map (which q entry DoF i looks at,
   * and its limits) is data in cfg so that the reference's off-by-one is reproducible. */
  if (lb && ub) {
    for (int i = 0; i < NV; ++i) {
      if (!c->use_bounds) { lb[i] = -1e30; ub[i] = 1e30; if (i >= nv) lb[i] = ub[i] = 0; continue; }
      if (i >= nv) { lb[i] = ub[i] = 0; continue; }
      const double qi = q[c->damper_qidx[i]], lo = c->damper_lo[i], hi = c->damper_hi[i], vm = c->damper_vmax[i];
      double l, u;
      if (qi <= lo + c->damper_qi) {
        l = -c->damper_coef * (qi - lo - c->damper_qs) / (c->damper_qi - c->damper_qs);
        if (l > vm) l = vm;
        if (l < -vm) l = -vm;
      } else l = -vm;
      if (qi >= hi - c->damper_qi) {
        u = c->damper_coef * (hi - qi - c->damper_qs) / (c->damper_qi - c->damper_qs);
        if (u < -vm) u = -vm;
        if (u > vm) u = vm;
      } else u = vm;
      if (l > 0) l = l * -1;                                      /* :621-625 */
      if (u < 0) u = u * -1;
      if (i >= c->lock_from) { l = 0; u = 0; }                    /* :627-630 */
      lb[i] = l; ub[i] = u;
    }
  }
}

/* ---------------------------------------------------------------- QP: Goldfarb–Idnani dual active set
 * min 1/2 x'Hx + g'x  s.t.  lb <= x <= ub, Clb <= Cx <= Cub   (the problem QP_Wrapper.py:45-48 hands qpOASES).
 * Constraint c in [0, n): bound on x_c; c in [n, n+p): row c-n of C. side 0: n'x >= lo (normal +a), side 1:
 * -a'x >= -hi. Equal lower/upper => equality (always active). |bound| >= 1e20 => absent (qpOASES INFTY). */
#define QP_INF 1e20

typedef struct {
  int n, p;
  const double *C, *lo_b, *hi_b, *lo_c, *hi_c;
} QpCons;

static double con_lo(const QpCons* Q, int c) { return c < Q->n ? (Q->lo_b ? Q->lo_b[c] : -1e30) : Q->lo_c[c - Q->n]; }
static double con_hi(const QpCons* Q, int c) { return c < Q->n ? (Q->hi_b ? Q->hi_b[c] : 1e30) : Q->hi_c[c - Q->n]; }
static void con_normal(const QpCons* Q, int c, int side, double* np) {
  const double s = side ? -1.0 : 1.0;
  if (c < Q->n) { memset(np, 0, sizeof(double) * Q->n); np[c] = s; }
  else for (int k = 0; k < Q->n; ++k) np[k] = s * Q->C[(size_t)(c - Q->n) * Q->n + k];
}
static double con_value(const QpCons* Q, int c, const double* x) { /* a'x */
  if (c < Q->n) return x[c];
  double s = 0;
  for (int k = 0; k < Q->n; ++k) s += Q->C[(size_t)(c - Q->n) * Q->n + k] * x[k];
  return s;
}

#define QN 32 /* max n handled by the oracle QP */

static int qp_solve_body(int n, int p, int m, const double* A, const double* bv, const double* H, const double* g, const double* C,
                         const double* lb, const double* ub, const double* Clb, const double* Cub, double* x, int* iters_out);
/* Iterative refinement at the final working set (QP_Wrapper.py:37 asks qpOASES for numRefinementSteps = 100; one step is what converges
 * here). 0 switches it off (tests measure what it buys). */
static int g_refine_steps = 1;
void orc_set_refine_steps(int k) { g_refine_steps = k < 0 ? 0 : k; }
int orc_get_refine_steps(void) { return g_refine_steps; }
/* Contract for a QP that was not solved (iteration cap, infeasible, numerical): x = 0. The reference ignores qpOASES'
 * return value; qpOASES' getPrimalSolution() does not write its argument unless the QP is solved, so xOpt keeps what it
 * held — zeros on the first QP (QP_Wrapper.py:50), the previous tick's answer afterwards (QP_Wrapper.py:71-73). The batched
 * path has no "previous answer" per call: it returns the first-call value, 0 (hold still), and the mirrors keep the stale
 * vector like the reference. */
int orc_qp_solve_ls(int n, int p, int m, const double* A, const double* bv, const double* H, const double* g, const double* C,
                    const double* lb, const double* ub, const double* Clb, const double* Cub, double* x, int* iters_out);
int orc_qp_solve(int n, int p, const double* H, const double* g, const double* C, const double* lb,
                 const double* ub, const double* Clb, const double* Cub, double* x, int* iters_out) {
  return orc_qp_solve_ls(n, p, 0, 0, 0, H, g, C, lb, ub, Clb, Cub, x, iters_out);
}
/* The same QP with its least-squares data: H = A'A, g = -A'b as QP_Wrapper.py:17-18 forms them, A (m x n) and b kept for the
 * refinement's residual (see qp_refine). */
int orc_qp_solve_ls(int n, int p, int m, const double* A, const double* bv, const double* H, const double* g, const double* C,
                    const double* lb, const double* ub, const double* Clb, const double* Cub, double* x, int* iters_out) {
  int st = -1;
  for (int i = 0; i < n; ++i) if ((lb && lb[i] != lb[i]) || (ub && ub[i] != ub[i])) st = WBC_QP_NUMERICAL;   /* NaN bound: refuse */
  for (int i = 0; i < p; ++i) if (Clb[i] != Clb[i] || Cub[i] != Cub[i]) st = WBC_QP_NUMERICAL;
  if (st < 0) st = qp_solve_body(n, p, m, A, bv, H, g, C, lb, ub, Clb, Cub, x, iters_out);
  else if (iters_out) *iters_out = 0;
  if (st == WBC_QP_OPTIMAL)
    for (int i = 0; i < n; ++i) if (!(fabs(x[i]) <= 1.7976931348623157e308)) st = WBC_QP_NUMERICAL;   /* NaN / Inf: never "optimal" */
  if (st != WBC_QP_OPTIMAL) for (int i = 0; i < n; ++i) x[i] = 0.0;
  return st;
}
/* One step of iterative refinement at the final working set W (normals n_k, right-hand sides b_k, multipliers u_k >= 0 with
 * grad f(x) = sum u_k n_k at the optimum). QP_Wrapper.py:37 sets qpOASES' numRefinementSteps = 100; this is its analogue for the
 * dual method's factors J = L^-T Q (J J' = H^-1; J' n_k = column k of [R; 0]):
 *     r1 = -(grad f(x) - sum u_k n_k),   r2_k = b_k - n_k'x,
 *     dx = J1 R^-T r2 + J2 J2' r1        (the KKT correction H dx - N'du = r1, N dx = r2 solved through the factors).
 * The residual comes from the UNFACTORED data. With the least-squares data at hand (the tick: H = A'A, g = -A'b, QP_Wrapper.py:17-18)
 * grad f = A'(A x - b) is formed as two products — it never sees the rounding of H. That is the point: on the benchmark tick the
 * posture rows weigh (0.001/26)^2 = 1.5e-9 against O(1) task rows (Robot_Wrapper4.py:1202-1204, 1437), so fl(A'A) carries that
 * block with 1e-5 relative error and the EXACT optimum of the rounded (H, g) already sits 1e-7 .. 1e-6 away from the exact optimum
 * of the least-squares problem (and from the exact optimum of any other rounding of H: numpy's, the device's). A residual from H
 * (the only choice for QP(H, g): orc_qp_solve) converges to the former and gains little; the least-squares residual converges to
 * the latter, which every implementation shares to ~cond(A) eps = 1e-11 (tests/test_oracle_qp.py measures both). */
static void qp_refine(int n, int q, int m, const double* A, const double* bv, const double* H, const double* g, const QpCons* Q,
                      double J[QN][QN], double R[QN][QN], const int* act, const int* act_side, const double* u, double* x) {
  double r1[QN], r2[QN], dy[QN], np[QN];
  if (A && m > 0) {                                   /* r1 = A'(b - A x) */
    for (int i = 0; i < n; ++i) r1[i] = 0;
    for (int r = 0; r < m; ++r) {
      double e = bv[r];
      for (int k = 0; k < n; ++k) e -= A[(size_t)r * n + k] * x[k];
      for (int k = 0; k < n; ++k) r1[k] += A[(size_t)r * n + k] * e;
    }
  } else {                                            /* r1 = -(H x + g) */
    for (int i = 0; i < n; ++i) { double s = g[i]; for (int k = 0; k < n; ++k) s += H[i * n + k] * x[k]; r1[i] = -s; }
  }
  for (int k = 0; k < q; ++k) {
    con_normal(Q, act[k], act_side[k], np);
    const double bk = act_side[k] ? -con_hi(Q, act[k]) : con_lo(Q, act[k]);
    double s = bk;
    for (int i = 0; i < n; ++i) { s -= np[i] * x[i]; r1[i] += u[k] * np[i]; }
    r2[k] = s;
  }
  for (int k = 0; k < q; ++k) {                       /* dy1 = R^-T r2 (R upper triangular: forward substitution on R') */
    double s = r2[k];
    for (int i = 0; i < k; ++i) s -= R[i][k] * dy[i];
    dy[k] = s / R[k][k];
  }
  for (int k = q; k < n; ++k) { double s = 0; for (int i = 0; i < n; ++i) s += J[i][k] * r1[i]; dy[k] = s; }   /* dy2 = J2' r1 */
  /* the correction is small against x (1e-6 on the tick, up to 1e-3 on a cond-1e10 problem); one that is not (> 0.25 max(1, |x|)) or is non-finite — a working set on the edge of dependence —
   * is not applied: x keeps the dual method's answer (same rule on the device) */
  double dx[QN], xmax = 1.0, dmax = 0.0;
  for (int i = 0; i < n; ++i) {
    double s = 0;
    for (int k = 0; k < n; ++k) s += J[i][k] * dy[k];
    dx[i] = s;
    if (fabs(x[i]) > xmax) xmax = fabs(x[i]);
    if (!(fabs(s) <= dmax)) dmax = fabs(s);       /* (a NaN sticks) */
  }
  if (!(dmax <= 0.25 * xmax)) return;
  for (int i = 0; i < n; ++i) x[i] += dx[i];
}

static int qp_solve_body(int n, int p, int m, const double* A, const double* bv, const double* H, const double* g, const double* C,
                         const double* lb, const double* ub, const double* Clb, const double* Cub, double* x, int* iters_out) {
  if (n > QN || p > 64) return WBC_QP_NUMERICAL;
  QpCons Q = {n, p, C, lb, ub, Clb, Cub};
  double L[QN][QN], J[QN][QN], R[QN][QN], d[QN], z[QN], r[QN], u[QN + 1], np[QN];
  int act[QN], act_side[QN], act_eq[QN], q = 0, iters = 0;
  const int ncon = n + p;
  char is_active[QN + 64];
  memset(is_active, 0, sizeof is_active);
  memset(R, 0, sizeof R);

  /* Cholesky H = L L' */
  for (int j = 0; j < n; ++j) {
    double s = H[j * n + j];
    for (int k = 0; k < j; ++k) s -= L[j][k] * L[j][k];
    if (!(s > 0)) { if (iters_out) *iters_out = 0; return WBC_QP_NUMERICAL; }
    L[j][j] = sqrt(s);
    for (int i = j + 1; i < n; ++i) {
      double t = H[i * n + j];
      for (int k = 0; k < j; ++k) t -= L[i][k] * L[j][k];
      L[i][j] = t / L[j][j];
    }
  }
  /* J = L^-T: column c of L^-1 by forward substitution, stored as row c of J */
  double jf2 = 0;
  for (int c = 0; c < n; ++c) {
    double y[QN];
    for (int i = 0; i < n; ++i) {
      double t = (i == c) ? 1.0 : 0.0;
      for (int k = c; k < i; ++k) t -= L[i][k] * y[k];
      y[i] = (i < c) ? 0.0 : t / L[i][i];
    }
    for (int i = 0; i < n; ++i) { J[c][i] = y[i]; jf2 += y[i] * y[i]; }
  }
  /* x = -H^-1 g = -J J' g */
  for (int k = 0; k < n; ++k) { double s = 0; for (int i = 0; i < n; ++i) s += J[i][k] * g[i]; d[k] = s; }
  for (int i = 0; i < n; ++i) { double s = 0; for (int k = 0; k < n; ++k) s += J[i][k] * d[k]; x[i] = -s; }

  const double eps2 = 2.220446049250313e-16 * 2.220446049250313e-16;
  int phase_eq = 1, eq_cursor = 0, status = WBC_QP_OPTIMAL;
  const int max_iter = 10 * (n + p) + 20;

  for (;;) {
    int ip = -1, ip_side = 0, ip_eq = 0;
    double s_ip = 0, b_ip = 0;
    if (phase_eq) { /* add every equality (lb == ub), in index order */
      while (eq_cursor < ncon) {
        const double lo = con_lo(&Q, eq_cursor), hi = con_hi(&Q, eq_cursor);
        if (lo == hi && fabs(lo) < QP_INF) break;
        ++eq_cursor;
      }
      if (eq_cursor < ncon) {
        ip = eq_cursor++; ip_side = 0; ip_eq = 1; b_ip = con_lo(&Q, ip);
        s_ip = con_value(&Q, ip, x) - b_ip;
      } else phase_eq = 0;
    }
    if (!phase_eq) { /* most violated inactive inequality */
      double worst = 0;
      for (int c = 0; c < ncon; ++c) {
        if (is_active[c]) continue;
        const double lo = con_lo(&Q, c), hi = con_hi(&Q, c);
        if (lo == hi && fabs(lo) < QP_INF) continue;
        const double v = con_value(&Q, c, x);
        if (lo > -QP_INF) {
          const double s = v - lo, tol = 1e-9 * fmax(1.0, fabs(lo));
          if (s < -tol && s < worst) { worst = s; ip = c; ip_side = 0; b_ip = lo; }
        }
        if (hi < QP_INF) {
          const double s = hi - v, tol = 1e-9 * fmax(1.0, fabs(hi));
          if (s < -tol && s < worst) { worst = s; ip = c; ip_side = 1; b_ip = -hi; }
        }
      }
      if (ip < 0) break; /* primal feasible: optimal */
      s_ip = worst;
    }
    con_normal(&Q, ip, ip_side, np);
    double np2 = 0;
    for (int k = 0; k < n; ++k) np2 += np[k] * np[k];
    double u_ip = 0;

    for (;;) { /* step towards satisfying constraint ip, dropping blocking constraints */
      if (++iters > max_iter) { status = WBC_QP_MAX_ITER; goto done; }
      for (int k = 0; k < n; ++k) { double s = 0; for (int i = 0; i < n; ++i) s += J[i][k] * np[i]; d[k] = s; }
      double zn = 0;
      for (int k = q; k < n; ++k) zn += d[k] * d[k];
      for (int i = 0; i < n; ++i) { double s = 0; for (int k = q; k < n; ++k) s += J[i][k] * d[k]; z[i] = s; }
      for (int i = q - 1; i >= 0; --i) { /* r = R^-1 d[0:q] */
        double s = d[i];
        for (int k = i + 1; k < q; ++k) s -= R[i][k] * r[k];
        r[i] = s / R[i][i];
      }
      const int have_step = zn > 100.0 * n * eps2 * jf2 * np2;
      double t1 = INFINITY; int l = -1;
      for (int k = 0; k < q; ++k)
        if (!act_eq[k] && r[k] > 0) { const double t = u[k] / r[k]; if (t < t1) { t1 = t; l = k; } }
      const double t2 = have_step ? -s_ip / zn : INFINITY;
      if (ip_eq && !have_step) { /* dependent equality */
        if (fabs(s_ip) <= 1e-9 * fmax(1.0, fabs(b_ip))) break; /* redundant: skip */
        status = WBC_QP_INFEASIBLE; goto done;
      }
      double t = ip_eq ? t2 : fmin(t1, t2);
      if (!(t < INFINITY)) { status = WBC_QP_INFEASIBLE; goto done; }
      if (!have_step) { /* dual step only */
        for (int k = 0; k < q; ++k) u[k] -= t * r[k];
        u_ip += t;
      } else {
        for (int i = 0; i < n; ++i) x[i] += t * z[i];
        for (int k = 0; k < q; ++k) u[k] -= t * r[k];
        u_ip += t;
      }
      if (have_step && t == t2) { /* full step: add ip to the active set (Givens on d[q..n-1]) */
        for (int j = n - 1; j > q; --j) {
          const double a = d[j - 1], bb = d[j];
          if (bb == 0) continue;
          const double h = hypot(a, bb), cc = a / h, ss = bb / h;
          d[j - 1] = h; d[j] = 0;
          for (int i = 0; i < n; ++i) {
            const double t1j = J[i][j - 1], t2j = J[i][j];
            J[i][j - 1] = cc * t1j + ss * t2j;
            J[i][j] = -ss * t1j + cc * t2j;
          }
        }
        for (int i = 0; i <= q; ++i) R[i][q] = d[i];
        u[q] = u_ip; act[q] = ip; act_side[q] = ip_side; act_eq[q] = ip_eq; is_active[ip] = 1;
        ++q;
        break;
      }
      /* partial step: drop blocking constraint l */
      is_active[act[l]] = 0;
      for (int k = l; k < q - 1; ++k) {
        act[k] = act[k + 1]; act_side[k] = act_side[k + 1]; act_eq[k] = act_eq[k + 1]; u[k] = u[k + 1];
        for (int i = 0; i < q; ++i) R[i][k] = R[i][k + 1];
      }
      for (int i = 0; i < q; ++i) R[i][q - 1] = 0;
      --q;
      for (int k = l; k < q; ++k) { /* re-triangularise: rotate rows k, k+1 of R and columns k, k+1 of J */
        const double a = R[k][k], bb = R[k + 1][k];
        if (bb == 0) continue;
        const double h = hypot(a, bb), cc = a / h, ss = bb / h;
        for (int jj = k; jj < q; ++jj) {
          const double t1j = R[k][jj], t2j = R[k + 1][jj];
          R[k][jj] = cc * t1j + ss * t2j;
          R[k + 1][jj] = -ss * t1j + cc * t2j;
        }
        R[k + 1][k] = 0;
        for (int i = 0; i < n; ++i) {
          const double t1j = J[i][k], t2j = J[i][k + 1];
          J[i][k] = cc * t1j + ss * t2j;
          J[i][k + 1] = -ss * t1j + cc * t2j;
        }
      }
      s_ip = (ip_side ? -1.0 : 1.0) * con_value(&Q, ip, x) - b_ip;
    }
  }
done:
  if (status == WBC_QP_OPTIMAL && A && m > 0)        /* (H, g) alone: a residual from H gains nothing where H is the rounding (measured): not refined, like the device */
    for (int k = 0; k < g_refine_steps; ++k) qp_refine(n, q, m, A, bv, H, g, &Q, J, R, act, act_side, u, x);
  if (iters_out) *iters_out = iters;
  return status;
}

/* KKT certificate of a candidate solution (independent of how it was found): returns
 * max(stationarity residual with the best non-negative multipliers is NOT computed here) — instead the
 * tests use scipy; this helper only reports primal infeasibility and objective. */
double orc_qp_objective(int n, const double* H, const double* g, const double* x) {
  double f = 0;
  for (int i = 0; i < n; ++i) {
    double s = 0;
    for (int k = 0; k < n; ++k) s += H[i * n + k] * x[k];
    f += x[i] * (0.5 * s + g[i]);
  }
  return f;
}


/* ---------------------------------------------------------------- manipulability-gradient posture target
 * qpJointb "MANI" (Robot_Wrapper4.py:1220-1242) and "HYBRID" (:1245-1260), restated LITERALLY:
 *  - the nq-sized configuration is perturbed at index i = the VELOCITY index of the DoF (so i = 3..6 touch the
 *    quaternion, and joint DoF i perturbs the angle of the joint before it);
 *  - the manipulability is sqrt(det(J J')) of pin.getJointJacobian(joint_id, LOCAL_WORLD_ALIGNED) with
 *    joint_id = 1 (i < 6) / i - 4 in MANI and joint_id = i - 6 in HYBRID (evaluated only when joint_id >= arm_base_id);
 *  - the perturbations accumulate: after each DoF q[i] is left at q[i] - deltaq;
 *  - updateState aliases current_joint_config to the perturbed array, so the configuration the rest of the tick sees
 *    (findConstraints, velDamperJointConstraints, integrate) is q_after.
 * mode 3 = MANI, 4 = HYBRID; literal = 0 gives the intended central difference (own q index, no accumulation, q restored).
 * u (nv) = the posture target before the (1/nv) w scaling; q_after (27) = configuration left behind. */
static double det6(double M[6][6]) { /* LU with partial pivoting, as numpy.linalg.det (LAPACK getrf) */
  double det = 1.0;
  for (int c = 0; c < 6; ++c) {
    int piv = c;
    for (int r = c + 1; r < 6; ++r) if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
    if (M[piv][c] == 0.0) return 0.0;
    if (piv != c) { for (int k = 0; k < 6; ++k) { double t = M[c][k]; M[c][k] = M[piv][k]; M[piv][k] = t; } det = -det; }
    det *= M[c][c];
    for (int r = c + 1; r < 6; ++r) {
      const double f = M[r][c] / M[c][c];
      for (int k = c + 1; k < 6; ++k) M[r][k] -= f * M[c][k];
    }
  }
  return det;
}
static double manipulability(const WbcModelBlob* m, const double* q, int joint_id) {
  double oMi[WBC_MAX_JOINTS * 12], J[6 * NV], Jj[6 * NV], G[6][6];
  orc_fk(m, q, oMi);
  orc_joint_jacobians(m, oMi, J);
  orc_frame_jacobian(m, oMi, J, -1, joint_id, 2, Jj);
  for (int a = 0; a < 6; ++a)
    for (int b = 0; b < 6; ++b) { double s = 0; for (int k = 0; k < m->nv; ++k) s += Jj[a * NV + k] * Jj[b * NV + k]; G[a][b] = s; }
  return sqrt(det6(G));
}
void orc_posture_target(const WbcModelBlob* m, const double* q_in, int mode, int arm_base_id, int literal,
                        double* u, double* q_after) {
  const double dq = 0.0002;
  double q[NQS];
  memcpy(q, q_in, sizeof q);
  const int nv = m->nv;
  for (int k = 0; k < nv; ++k) u[k] = (mode == 4) ? ((k < 6) ? q_in[k] : q_in[k + 1]) : 0.0;   /* HYBRID starts from "PREV" */
  for (int i = 0; i < nv; ++i) {
    int joint_id;
    if (mode == 3) joint_id = (i < 6) ? 1 : i + 1 - 5;
    else { joint_id = i - 6; if (joint_id < arm_base_id) continue; }
    const int qi = literal ? i : ((i < 6) ? i : i + 1);
    if (!literal && mode == 3 && i >= 6) joint_id = i - 4;
    q[qi] += dq;
    const double f1 = manipulability(m, q, joint_id);
    q[qi] -= 2 * dq;
    const double f2 = manipulability(m, q, joint_id);
    u[i] = 0.5 * (f1 - f2) / dq;
    if (!literal) q[qi] = q_in[qi];
  }
  if (q_after) memcpy(q_after, q, sizeof q);
}
void orc_posture_batch(const WbcModelBlob* const* models, int B, const double* q, const int32_t* model_id, int mode,
                       const int32_t* arm_base_id, int literal, double* u, double* q_after, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int b = 0; b < B; ++b) {
    const int mi = model_id ? model_id[b] : 0;
    double ub[NV], qa[NQS];
    memset(ub, 0, sizeof ub);
    orc_posture_target(models[mi], q + (size_t)b * NQS, mode, arm_base_id[mi], literal, ub, qa);
    memcpy(u + (size_t)b * NV, ub, sizeof ub);
    if (q_after) memcpy(q_after + (size_t)b * NQS, qa, sizeof qa);
  }
}


/* ---------------------------------------------------------------- closing the loop: the tail of runWBC
 * updateState(joint_config, base_config, running=True) (Robot_Wrapper4.py:1397-1399, 387-428): the new configuration is
 * [current base xyz, base quaternion handed in (IMU), joints of q_next]; FK; then trunkWorldPos (:1297-1327) re-estimates
 * the base position from the four stance-foot TARGETS: trunk_pos = WPA - WRB . BPA with WPA the mean foot target and BPA
 * the mean world-frame offset foot - trunk (rotated by WRB once more, as the reference does). The second FK of
 * updateState only refreshes robot_data; the next tick's FK does that here.
 * foot_targets = ee_target of the tick ([5][3], first four used: FR, FL, RR, RL). imu may be NULL (q_next's quaternion). */
void orc_update_state(const WbcModelBlob* m, const double* q_cur, const double* q_next, const double* imu,
                      const double* foot_targets, double* q_new) {
  double cfg[NQS], oMi[WBC_MAX_JOINTS * 12], Mt[12], Mf[12];
  memset(cfg, 0, sizeof cfg);
  for (int i = 0; i < 3; ++i) cfg[i] = q_cur[i];
  for (int i = 0; i < 4; ++i) cfg[3 + i] = imu ? imu[i] : q_next[3 + i];
  for (int i = 7; i < m->nq; ++i) cfg[i] = q_next[i];
  orc_fk(m, cfg, oMi);
  orc_frame_placement(m, oMi, WBC_FR_TRUNK, Mt);
  double bpa[4][3];
  for (int e = 0; e < 4; ++e) {
    orc_frame_placement(m, oMi, WBC_FR_EE0 + e, Mf);
    for (int i = 0; i < 3; ++i) bpa[e][i] = Mf[9 + i] - Mt[9 + i];
  }
  double WPA[3], BPA[3];
  for (int i = 0; i < 3; ++i) {
    WPA[i] = (foot_targets[0 + i] + foot_targets[3 + i] + foot_targets[6 + i] + foot_targets[9 + i]) / 4;   /* :1321 */
    BPA[i] = (bpa[0][i] + bpa[1][i] + bpa[2][i] + bpa[3][i]) / 4;                                              /* :1323 */
  }
  double rb[3];
  m3_vec(Mt, BPA, rb);
  memcpy(q_new, cfg, sizeof cfg);
  for (int i = 0; i < 3; ++i) q_new[i] = WPA[i] - rb[i];                                                       /* :1325 */
}
void orc_update_state_batch(const WbcModelBlob* const* models, int B, const double* q_cur, const double* q_next,
                            const double* imu, const double* foot_targets, const int32_t* model_id, double* q_new) {
  for (int b = 0; b < B; ++b)
    orc_update_state(models[model_id ? model_id[b] : 0], q_cur + (size_t)b * NQS, q_next + (size_t)b * NQS,
                     imu ? imu + (size_t)b * 4 : 0, foot_targets + (size_t)b * 15, q_new + (size_t)b * NQS);
}

/* ---------------------------------------------------------------- batched drivers (OpenMP) */

/* one runWBC tick per instance: assemble -> QP -> integrate (Robot_Wrapper4.py:1348-1397). cfgs[model]. */
void orc_tick_batch(const WbcModelBlob* const* models, const WbcConfig* cfgs, int B, const WbcTickIn* in, double dt,
                    const WbcTickOut* out, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 16)
  for (int b = 0; b < B; ++b) {
    const int mi = in->model_id ? in->model_id[b] : 0;
    const WbcModelBlob* m = models[mi];
    const WbcConfig* c = cfgs + mi;
    const int p = orc_constraint_rows(c);
    double C[WBC_MAX_P * NV], Clb[WBC_MAX_P], Cub[WBC_MAX_P], lb[NV], ub[NV], H[NV * NV], g[NV], x[NV];
    double At[WBC_MAX_M * NV], bt[WBC_MAX_M];          /* the task stack itself: the refinement's residual (qp_refine) */
    int it = 0;
    double qcon[NQS];
    orc_assemble_one(m, c, in, b, dt, At, bt, C, Clb, Cub, lb, ub, H, g, qcon);
    const int st = orc_qp_solve_ls(NV, p, orc_task_rows(c), At, bt, H, g, C, lb, ub, Clb, Cub, x, &it);
    if (out->qdot) memcpy(out->qdot + (size_t)b * NV, x, sizeof x);
    if (out->status) out->status[b] = st;
    if (out->iters) out->iters[b] = it;
    if (out->q_next) {
      double v[NV], qn[NQS];
      for (int k = 0; k < NV; ++k) v[k] = x[k] * dt;
      memset(qn, 0, sizeof qn);
      orc_integrate(m, qcon, v, qn);   /* jointVelocitiestoConfig integrates current_joint_config (:441) */
      memcpy(out->q_next + (size_t)b * NQS, qn, sizeof qn);
    }
  }
}

void orc_assemble_batch(const WbcModelBlob* const* models, const WbcConfig* cfgs, int B, const WbcTickIn* in, double dt,
                        const WbcQpData* o, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int b = 0; b < B; ++b) {
    const int mi = in->model_id ? in->model_id[b] : 0;
    const WbcConfig* c = cfgs + mi;
    const int m = orc_task_rows(c), p = orc_constraint_rows(c);
    orc_assemble_one(models[mi], c, in, b, dt, o->A ? o->A + (size_t)b * m * NV : 0, o->b ? o->b + (size_t)b * m : 0,
                     o->C ? o->C + (size_t)b * p * NV : 0, o->Clb ? o->Clb + (size_t)b * p : 0,
                     o->Cub ? o->Cub + (size_t)b * p : 0, o->lb ? o->lb + (size_t)b * NV : 0,
                     o->ub ? o->ub + (size_t)b * NV : 0, o->H ? o->H + (size_t)b * NV * NV : 0,
                     o->g ? o->g + (size_t)b * NV : 0, 0);
  }
}

/* QP(A, b, ...) as QP_Wrapper.py:10-53 takes it: H = A'A and g = -A'b formed here as numpy forms them (:17-18) */
void orc_qp_ls_batch(int B, int n, int p, int m, const double* A, const double* bv, const double* C, const double* lb,
                     const double* ub, const double* Clb, const double* Cub, double* x, int32_t* status, int32_t* iters,
                     int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 16)
  for (int b = 0; b < B; ++b) {
    double H[QN * QN], g[QN];
    const double* Ab = A + (size_t)b * m * n;
    const double* bb = bv + (size_t)b * m;
    int it = 0, st = WBC_QP_NUMERICAL;
    if (n <= QN) {
      for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) { double s = 0; for (int r = 0; r < m; ++r) s += Ab[(size_t)r * n + i] * Ab[(size_t)r * n + j]; H[i * n + j] = s; }
        double s = 0; for (int r = 0; r < m; ++r) s += Ab[(size_t)r * n + i] * bb[r];
        g[i] = -s;
      }
      st = orc_qp_solve_ls(n, p, m, Ab, bb, H, g, C ? C + (size_t)b * p * n : 0, lb ? lb + (size_t)b * n : 0, ub ? ub + (size_t)b * n : 0,
                           Clb ? Clb + (size_t)b * p : 0, Cub ? Cub + (size_t)b * p : 0, x + (size_t)b * n, &it);
    }
    if (status) status[b] = st;
    if (iters) iters[b] = it;
  }
}

void orc_qp_batch(int B, int n, int p, const double* H, const double* g, const double* C, const double* lb,
                  const double* ub, const double* Clb, const double* Cub, double* x, int32_t* status, int32_t* iters,
                  int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 16)
  for (int b = 0; b < B; ++b) {
    int it = 0;
    const int st = orc_qp_solve(n, p, H + (size_t)b * n * n, g + (size_t)b * n, C ? C + (size_t)b * p * n : 0,
                                lb ? lb + (size_t)b * n : 0, ub ? ub + (size_t)b * n : 0, Clb ? Clb + (size_t)b * p : 0,
                                Cub ? Cub + (size_t)b * p : 0, x + (size_t)b * n, &it);
    if (status) status[b] = st;
    if (iters) iters[b] = it;
  }
}

/* FK + Jacobians for a batch (outputs as WbcFkOut). */
/* nj_stride / nf_stride: joints / frames per instance in o->oMi / o->oMf (the largest model of a mixed batch, include/wbc.h
 * WbcFkOut); rows beyond an instance's own model are left as the caller initialised them (zeros). */
void orc_fk_batch(const WbcModelBlob* const* models, int B, const double* q, const int32_t* model_id,
                  const WbcFkOut* o, int nthreads, int nj_stride, int nf_stride) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int b = 0; b < B; ++b) {
    const WbcModelBlob* m = models[model_id ? model_id[b] : 0];
    double oMi[WBC_MAX_JOINTS * 12], J[6 * NV], com[3], Jcom[3 * NV];
    orc_fk(m, q + (size_t)b * NQS, oMi);
    orc_joint_jacobians(m, oMi, J);
    if (o->oMi) memcpy(o->oMi + (size_t)b * nj_stride * 12, oMi, sizeof(double) * m->njoints * 12);
    if (o->oMf) for (int f = 0; f < m->nframes; ++f) orc_frame_placement(m, oMi, f, o->oMf + ((size_t)b * nf_stride + f) * 12);
    if (o->J) memcpy(o->J + (size_t)b * 6 * NV, J, sizeof J);
    if (o->com || o->Jcom) {
      orc_com(m, oMi, J, com, Jcom);
      if (o->com) memcpy(o->com + (size_t)b * 3, com, sizeof com);
      if (o->Jcom) memcpy(o->Jcom + (size_t)b * 3 * NV, Jcom, sizeof Jcom);
    }
  }
}

void orc_integrate_batch(const WbcModelBlob* const* models, int B, const double* q, const double* v,
                         const int32_t* model_id, double dt, double* qn) {
  for (int b = 0; b < B; ++b) {
    const WbcModelBlob* m = models[model_id ? model_id[b] : 0];
    double vv[NV];
    for (int k = 0; k < NV; ++k) vv[k] = v[(size_t)b * NV + k] * dt;
    memset(qn + (size_t)b * NQS, 0, sizeof(double) * NQS);
    orc_integrate(m, q + (size_t)b * NQS, vv, qn + (size_t)b * NQS);
  }
}
