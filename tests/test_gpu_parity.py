"""GPU parity tests: the HIP path, called through the C-ABI (ctypes), against the CPU oracle on identical inputs.

Tolerances: fp64 throughout. Kinematics / assembly agree to rounding (1e-11 relative); the QP solution is compared at
the 1e-5 max-abs bound BASELINE.json states for q̇ (cond(H) ~ 3e9 makes ~1e-7 the realistic floor between two
different but exact solvers: textbook Givens/R on the CPU vs Householder/R^-1 on the wavefront).
"""
import numpy as np
import pytest

import common
import oracle
import wbc_capi as capi
import wbc_model
from wbc_batch import WbcBatch

pytestmark = pytest.mark.gpu
DT = 0.002
QDOT_TOL = 1e-5          # north_star: "within 1e-5 max-abs on identical inputs"
REFINED_TOL = 1e-7       # what a path with the iterative refinement on (option refine, default 1) is held to against the oracle, which refines too


def relerr(a, b):
    if a.size == 0:
        return 0.0
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


@pytest.fixture(scope="module")
def wx200():
    return wbc_model.load_model("a1_wx200")


@pytest.fixture(scope="module")
def px100():
    return wbc_model.load_model("a1_px100_pin_ver")


def test_library_reports_gfx950_build():
    lib = capi.load_library()
    assert b"gfx950" in lib.wbc_version()


@pytest.mark.parametrize("name", ["a1_wx200", "a1_px100_pin_ver"])
def test_fk_jacobians_parity(name):
    m = wbc_model.load_model(name)
    rng = np.random.default_rng(3)
    import wbc_workload
    q = wbc_workload.sample_q(m, 300, rng)
    q[0] = m.neutral()
    ref = oracle.fk([m], q)
    bt = WbcBatch(m, 512)
    got = bt.fk(q)
    for k in ("oMi", "oMf", "J", "com", "Jcom"):
        assert np.abs(got[k] - ref[k]).max() < 1e-12, k
    bt.close()


@pytest.mark.parametrize("cfg_name,with_rot", [("c1", False), ("c2", False), ("full", True), ("everything", True)])
def test_assemble_parity(wx200, cfg_name, with_rot):
    cfg = common.config(cfg_name, wx200)
    B = 200
    d = common.tick_inputs(wx200, cfg, B, seed=11, with_rot=with_rot)
    ref = oracle.assemble([wx200], [cfg], d, DT, B)
    bt = WbcBatch(wx200, B)
    bt.configure(cfg)
    assert bt.task_rows == ref["A"].shape[1] and bt.constraint_rows == ref["C"].shape[1]
    got = bt.assemble(d, DT)
    for k in ("A", "b", "H", "g", "C", "Clb", "Cub", "lb", "ub"):
        assert got[k].shape == ref[k].shape, k
        assert relerr(got[k], ref[k]) < 1e-11, (k, relerr(got[k], ref[k]))
    bt.close()


def _random_qps(rng, B, n, p, n_eq, fixed_value=0.0):
    A = rng.normal(size=(B, n + 6, n))
    H = np.einsum("bmi,bmj->bij", A, A) + 1e-3 * np.eye(n)
    g = rng.normal(size=(B, n)) * 4
    C = rng.normal(size=(B, p, n))
    lb, ub = -rng.uniform(0.02, 0.8, (B, n)), rng.uniform(0.02, 0.8, (B, n))
    lb[:, -2:] = fixed_value          # fixed variables (lb == ub): the kernel presolves them out
    ub[:, -2:] = fixed_value
    cl, cu = -rng.uniform(0.02, 0.8, (B, p)), rng.uniform(0.02, 0.8, (B, p))
    cl[:, :n_eq] = cu[:, :n_eq] = rng.normal(size=(B, n_eq)) * 0.1
    return H, g, C, lb, ub, cl, cu


@pytest.mark.parametrize("n,p,n_eq", [(26, 16, 12), (25, 10, 4), (12, 6, 2), (26, 0, 0), (3, 2, 0)])
def test_qp_parity_random(wx200, n, p, n_eq):
    rng = np.random.default_rng(100 + n)
    B = 300
    H, g, C, lb, ub, cl, cu = _random_qps(rng, B, n, p, n_eq)
    bt = WbcBatch(wx200, B)
    if p:
        x, st, it = bt.qp_solve(H, g, C, lb, ub, cl, cu)
        xr, sr, ir = oracle.qp_solve(H, g, C, lb, ub, cl, cu)
    else:
        x, st, it = bt.qp_solve(H, g, None, lb, ub)
        xr, sr, ir = oracle.qp_solve(H, g, None, lb, ub)
    assert (st == sr).all(), np.nonzero(st != sr)
    ok = sr == 0
    assert ok.sum() > B // 2
    assert np.abs(x[ok] - xr[ok]).max() < 1e-9
    assert np.abs(it[ok] - ir[ok]).max() == 0        # same working-set changes as the textbook method
    # non-zero fixed values: their contribution moves into g and the row bounds
    H, g, C, lb, ub, cl, cu = _random_qps(rng, 64, n, p, n_eq, fixed_value=0.07)
    x, st, it = bt.qp_solve(H, g, C if p else None, lb, ub, cl if p else None, cu if p else None)
    xr, sr, ir = oracle.qp_solve(H, g, C if p else None, lb, ub, cl if p else None, cu if p else None)
    assert (st == sr).all()
    ok = sr == 0
    assert np.abs(x[ok] - xr[ok]).max() < 1e-9 and np.abs(x[ok][:, -2:] - 0.07).max() < 1e-12
    bt.close()


@pytest.mark.parametrize("m,n,p,lanes,B", [(18, 14, 6, 4, 301), (40, 26, 16, 2, 301), (12, 8, 0, 4, 64), (40, 20, 20, 2, 65), (30, 20, 8, 2, 65), (24, 16, 6, 4, 130)])
def test_qp_ls_packed_refined_matches_oracle(wx200, m, n, p, lanes, B):
    """QP(A, b, ...) on the packed kernel (csrc/wbc_k_qpp.hip: four problems per wavefront for n, p <= 16, else two; B not a multiple of either),
    ill-conditioned least-squares data (posture-like rows of 3e-5 under O(1) rows: the refinement has something to repair), rows active on both
    sides, one equality row: same status and iteration count as the oracle, refined answers within 1e-8 of the oracle's refined answers, and the
    one-per-wavefront kernel (option packed_kernel 0) agrees. The plain answer is off by more than 1e-6: the step is what closes the gap."""
    rng = np.random.default_rng(40 + n + p)
    A = np.concatenate([rng.normal(size=(B, m - n, n)), np.broadcast_to(3e-5 * np.eye(n), (B, n, n))], axis=1)
    b = np.concatenate([rng.normal(size=(B, m - n)), 3e-5 * rng.normal(size=(B, n))], axis=1)
    C = rng.normal(size=(B, p, n)) if p else None
    lb, ub = -rng.uniform(0.5, 2.0, (B, n)), rng.uniform(0.5, 2.0, (B, n))
    lb[:, -1] = ub[:, -1] = 0.03                     # one fixed variable
    cl, cu = (-rng.uniform(0.1, 1.0, (B, p)), rng.uniform(0.1, 1.0, (B, p))) if p else (None, None)
    if p:
        cl[:, 0] = cu[:, 0] = 0.05                   # one equality row
    xr, sr, ir = oracle.qp_solve_ls(A, b, C, lb, ub, cl, cu)
    ok = sr == 0
    assert ok.mean() > 0.9
    bt = WbcBatch(wx200, B)
    x, st, it = bt.qp_solve_ls(A, b, C, lb, ub, cl, cu)
    assert bt.stat("last_qp_path") == lanes
    assert (st == sr).all() and (it[ok] == ir[ok]).all()
    assert np.abs(x - xr)[ok].max() < 1e-8, np.abs(x - xr)[ok].max()
    bt.set_option("refine", 0)
    x0, st0, _ = bt.qp_solve_ls(A, b, C, lb, ub, cl, cu)
    assert (st0 == sr).all() and (2 * p > n or np.abs(x0 - xr)[ok].max() > 1e-6)      # (a vertex solution leaves the objective nothing to decide)
    bt.set_option("refine", 1)
    bt.set_option("packed_kernel", 0)
    x1, st1, it1 = bt.qp_solve_ls(A, b, C, lb, ub, cl, cu)
    assert bt.stat("last_qp_path") == 1
    assert (st1 == sr).all() and (it1[ok] == ir[ok]).all() and np.abs(x1 - xr)[ok].max() < 1e-8
    bt.close()


def test_qp_ls_forms_H_and_g(wx200):
    """QP(A, b, ...) boundary: H = A'A and g = -A'b formed on the device equal numpy's (QP_Wrapper.py:17-18)."""
    rng = np.random.default_rng(5)
    B, m, n, p = 64, 32, 26, 16
    A = rng.normal(size=(B, m, n))
    A[:, 6:, :] = 0
    A[:, 6:, :] += np.eye(n)[None] * 0.03
    b = rng.normal(size=(B, m))
    C = rng.normal(size=(B, p, n))
    lb, ub = -np.ones((B, n)), np.ones((B, n))
    cl, cu = -np.ones((B, p)) * 0.3, np.ones((B, p)) * 0.3
    bt = WbcBatch(wx200, B)
    for mfma in (False, True):
        x, st, it, Ho, go = bt.qp_solve_ls(A, b, C, lb, ub, cl, cu, use_mfma=mfma, want_Hg=True)
        Hr = np.einsum("bmi,bmj->bij", A, A)
        gr = -np.einsum("bmi,bm->bi", A, b)
        assert relerr(Ho, Hr) < 1e-13 and relerr(go, gr) < 1e-13
        xr, sr, _ = oracle.qp_solve(Hr, gr, C, lb, ub, cl, cu)
        assert (st == sr).all()
        assert np.abs(x - xr)[sr == 0].max() < 1e-7
    bt.close()


@pytest.mark.parametrize("cfg_name,B,with_rot", [("c1", 1, False), ("c2", 1024, False), ("c3", 4096, False),
                                                 ("full", 512, True), ("everything", 512, True)])
def test_tick_parity(wx200, cfg_name, B, with_rot):
    cfg = common.config(cfg_name, wx200)
    d = common.tick_inputs(wx200, cfg, B, seed=21, with_rot=with_rot)
    ref = oracle.tick([wx200], [cfg], d, DT, B, nthreads=8)
    bt = WbcBatch(wx200, B)
    bt.configure(cfg)
    got = bt.tick(d, DT, want_q_next=True)
    assert (got["status"] == ref["status"]).all()
    ok = ref["status"] == 0
    assert ok.mean() > 0.9
    err = np.abs(got["qdot"] - ref["qdot"])[ok].max()
    print("%s: qdot max-abs err %.3e, iters gpu mean %.2f / oracle %.2f" % (cfg_name, err, got["iters"].mean(), ref["iters"].mean()))
    assert err < QDOT_TOL
    assert np.abs(got["q_next"] - ref["q_next"])[ok].max() < 1e-7
    bt.close()


@pytest.mark.parametrize("mode,literal", [("HYBRID", True), ("HYBRID", False), ("MANI", True), ("MANI", False)])
def test_posture_target_parity(wx200, px100, mode, literal):
    """qpJointb MANI / HYBRID (Robot_Wrapper4.py:1220-1260): u and the configuration left behind, both morphologies."""
    import wbc_workload
    B = 256
    models = [wx200, px100]
    cfgs = [wbc_model.sim3_config(m, Joint=mode, posture_literal=literal) for m in models]
    rng = np.random.default_rng(17)
    mid = (np.arange(B) % 2).astype(np.int32)
    qs = [wbc_workload.sample_q(m, B, rng) for m in models]
    q = np.where(mid[:, None] == 0, qs[0], qs[1])
    ur, qar = oracle.posture_target(models, cfgs, q, mid, nthreads=8)
    bt = WbcBatch(models, B)
    for i, c in enumerate(cfgs):
        bt.configure(c, i)
    u, qa = bt.posture_target(q, mid)
    assert bt.stat("last_posture_par") == 2       # every sweep on a lane of its own, three instances per wavefront (wbc_posture_par3_kernel: <= 21 sweeps)
    # f = sqrt(det(J J')) is O(1..10) and is differenced over 2e-4: rounding in f (1e-16 relative) shows as ~1e-11 in u
    assert np.abs(u - ur).max() < 1e-9
    assert (qa == qar).all()                      # same IEEE operations on q: bit-equal
    bt.set_option("posture_par", 3)               # one instance per wavefront, every finite-difference POINT on a lane of its own (wbc_posture_par_kernel):
    u3, qa3 = bt.posture_target(q[:B - 3], mid[:B - 3])     # the same arithmetic per evaluation — bit for bit (and a batch that is not a multiple of three)
    assert bt.stat("last_posture_par") == 1
    u2, qa2 = (bt.set_option("posture_par", 1), bt.posture_target(q[:B - 3], mid[:B - 3]))[1]
    assert (u3 == u2).all() and (qa3 == qa2).all() and (u2 == u[:B - 3]).all()
    bt.set_option("posture_par", 0)               # the sequential whole-tree kernel (52 sweeps per instance) says the same
    u1, qa1 = bt.posture_target(q, mid)
    assert bt.stat("last_posture_par") == 0
    assert np.abs(u1 - ur).max() < 1e-9 and (qa1 == qar).all() and np.abs(u1 - u).max() < 1e-9
    assert ((u == 0) == (u1 == 0)).all()          # the same DoF are exactly zero (sweeps that cannot change f, DoF the loop skips)
    assert np.abs(ur).max() > 1e-3
    if literal:
        assert np.abs(qa - q).max() == pytest.approx(2e-4, rel=1e-6)
    else:
        assert (qa == q).all()
    bt.close()


@pytest.mark.parametrize("cfg_name,B", [("c3_hybrid", 1024), ("c3_hybrid_clean", 512), ("c3_mani", 512), ("hybrid_grip_com", 512)])
def test_tick_parity_posture_modes(wx200, cfg_name, B):
    """A whole tick under sim3.py's own posture mode ("HYBRID", sim3.py:145) and "MANI": wbc_tick runs the posture
    kernel itself, and in literal mode the rest of the tick sees the perturbed configuration."""
    cfg = common.config(cfg_name, wx200)
    d = common.tick_inputs(wx200, cfg, B, seed=23)
    ref = oracle.tick([wx200], [cfg], d, DT, B, nthreads=8)
    bt = WbcBatch(wx200, B)
    bt.configure(cfg)
    got = bt.tick(d, DT, want_q_next=True)
    # HYBRID as sim3.py sets it: the packed kernel forms the (static) target itself; MANI: wbc_posture_par_kernel + the packed kernel's QCON variant
    # (tasks at q, constraints / bounds / integration at the perturbed state); Grip contact + CoM box: the general kernel
    assert bt.stat("last_path") == (0 if cfg_name == "hybrid_grip_com" else 2)
    assert (got["status"] == ref["status"]).all()
    ok = ref["status"] == 0
    assert ok.mean() > 0.9
    err = np.abs(got["qdot"] - ref["qdot"])[ok].max()
    print("%s: qdot max-abs err %.3e" % (cfg_name, err))
    if cfg_name == "c3_mani":                       # ... and the one-instance compact kernel agrees
        bt.set_option("packed_kernel", 0)
        bt.set_option("refine", 0)                  # (the compact kernel does not refine: with the refinement on these ticks take the general kernel)
        one = bt.tick(d, DT, want_q_next=True)
        assert bt.stat("last_path") == 1 and (one["status"] == ref["status"]).all()
        assert np.abs(one["qdot"] - got["qdot"])[ok].max() < QDOT_TOL and np.abs(one["q_next"] - got["q_next"])[ok].max() < 1e-7   # (q_next = q + qd dt)
        bt.set_option("refine", 1)
        gen = bt.tick(d, DT, want_q_next=True)
        assert bt.stat("last_path") == 0 and (gen["status"] == ref["status"]).all()
        assert np.abs(gen["qdot"] - ref["qdot"])[ok].max() < REFINED_TOL
        bt.set_option("packed_kernel", 1)
    assert err < REFINED_TOL
    assert np.abs(got["q_next"] - ref["q_next"])[ok].max() < 1e-7
    a, ar = bt.assemble(d, DT), oracle.assemble([wx200], [cfg], d, DT, B)
    for k in ("A", "b", "H", "g", "C", "Clb", "Cub", "lb", "ub"):
        assert relerr(a[k], ar[k]) < 1e-9, (k, relerr(a[k], ar[k]))
    if cfg.posture_literal:
        # the leak is visible: integrating from q instead of the perturbed state would be off by the 2e-4 perturbation
        plain = bt.integrate(d["q"], got["qdot"], DT)
        assert np.abs(plain - got["q_next"])[ok].max() > 1e-4
    bt.close()


def test_tick_custom_posture_and_q_con(wx200):
    """WBC_JOINT_CUSTOM: u supplied per instance; q_con: a second configuration for constraints / bounds / integrate."""
    B = 256
    cfg = common.config("c3_custom", wx200)
    d = common.tick_inputs(wx200, cfg, B, seed=29)
    rng = np.random.default_rng(4)
    d["posture_u"] = rng.normal(size=(B, 26))
    d["q_con"] = d["q"].copy()
    d["q_con"][:, 7:] += rng.normal(0, 1e-3, (B, 20))
    ref = oracle.tick([wx200], [cfg], d, DT, B, nthreads=8)
    bt = WbcBatch(wx200, B)
    bt.configure(cfg)
    got = bt.tick(d, DT, want_q_next=True)
    assert (got["status"] == ref["status"]).all()
    ok = ref["status"] == 0
    assert np.abs(got["qdot"] - ref["qdot"])[ok].max() < QDOT_TOL
    assert np.abs(got["q_next"] - ref["q_next"])[ok].max() < 1e-7
    del d["posture_u"]
    with pytest.raises(capi.WbcError):
        bt.tick(d, DT)
    bt.close()


@pytest.mark.parametrize("cfg_name", ["c3", "c2", "everything", "hybrid_grip_com", "c3_nobounds", "c3_two_feet", "c3_trunk_task"])
def test_contact_presolve_and_general_path_agree(wx200, px100, cfg_name):
    """The structural elimination of the stance-foot equalities (default) and the general path (option presolve = 0)
    solve the same QP: both within tolerance of the oracle, same status, on both morphologies (n' = 14 and 13)."""
    B = 768
    models = [wx200, px100]
    cfgs = [common.config(cfg_name, m) for m in models]
    mid = (np.arange(B) % 2).astype(np.int32)
    parts = [common.tick_inputs(m, c, B, seed=51 + i, with_rot=(cfg_name in ("everything", "c3_trunk_task")))
             for i, (m, c) in enumerate(zip(models, cfgs))]
    if cfg_name == "c3_trunk_task":          # a MOVING trunk reference (omega_ref != 0); the packed kernel honours the gripper's orientation reference too
        for prt in parts:
            prt["trunk_prev_rot"] = prt["trunk_prev_rot"] + 0.0     # (with_rot set them: R*_prev != R*)
    d = {k: np.where(mid.reshape((B,) + (1,) * (parts[0][k].ndim - 1)) == 0, parts[0][k], parts[1][k]) for k in parts[0]}
    d["model_id"] = mid
    ref = oracle.tick(models, cfgs, d, DT, B, nthreads=8)
    bt = WbcBatch(models, B)
    for i, c in enumerate(cfgs):
        bt.configure(c, i)
    res = {}
    # (presolve, sim3_kernel, packed_kernel): packed compact kernel (four instances per wavefront; where the configuration is of
    # the sim3 family, else the same as the next) / compact sim3 kernel / general kernel with presolve / general path only
    for key in ((1, 1, 1), (1, 1, 0), (1, 0, 0), (0, 0, 0)):
        bt.set_option("presolve", key[0])
        bt.set_option("sim3_kernel", key[1])
        bt.set_option("packed_kernel", key[2])
        bt.set_option("refine", 0 if key == (1, 1, 0) else 1)      # the compact kernel is only chosen with the refinement off (it has no room for it)
        res[key] = bt.tick(d, DT, want_q_next=True)
        assert (res[key]["status"] == ref["status"]).all(), key
        if key == (1, 1, 1):
            packed_ran = bt.stat("last_path") == 2
        if key == (1, 1, 0):
            assert bt.stat("last_path") == (1 if cfg_name in ("c3", "c3_nobounds") else 0), (cfg_name, bt.stat("last_path"))   # (orientation references: packed or general)
    ok = ref["status"] == 0
    assert ok.mean() > 0.9
    errs = {k: np.abs(v["qdot"] - ref["qdot"])[ok].max() for k, v in res.items()}
    print("%s: packed (%s) err %.3e, sim3 kernel err %.3e, presolve err %.3e, general err %.3e, iters %.2f / %.2f / %.2f / %.2f / oracle %.2f" % (
        cfg_name, "ran" if packed_ran else "not eligible", errs[(1, 1, 1)], errs[(1, 1, 0)], errs[(1, 0, 0)], errs[(0, 0, 0)],
        res[(1, 1, 1)]["iters"][ok].mean(), res[(1, 1, 0)]["iters"][ok].mean(), res[(1, 0, 0)]["iters"][ok].mean(),
        res[(0, 0, 0)]["iters"][ok].mean(), ref["iters"][ok].mean()))
    assert packed_ran == (cfg_name in ("c3", "c3_trunk_task"))        # (the trunk task: the packed kernel's TRUNK variant)
    assert max(errs.values()) < QDOT_TOL
    # every refining path lands on the exact least-squares optimum like the oracle: 1e-7 apart at most (1e-6 without, cond(H) ~ 3e9: the compact kernel)
    assert max(v for k, v in errs.items() if k != (1, 1, 0)) < REFINED_TOL, errs
    for key in ((1, 1, 1), (1, 1, 0)):
        assert np.abs(res[key]["qdot"] - res[(0, 0, 0)]["qdot"])[ok].max() < QDOT_TOL
        assert np.abs(res[key]["q_next"] - ref["q_next"])[ok].max() < 1e-7
    if cfg_name == "c3":
        assert np.abs(res[(1, 1, 0)]["qdot"] - res[(1, 0, 0)]["qdot"])[ok].max() < QDOT_TOL  # same reduced QP on both one-instance kernels (one of them refined)
        assert (res[(1, 1, 1)]["iters"] == res[(1, 1, 0)]["iters"])[ok].mean() > 0.98        # and the same working-set changes when packed
    bt.close()


def test_packed_kernel_honours_the_grippers_orientation_reference(wx200, px100):
    """The reference's calcTargetVelEE3 always carries orientation references (RW4:1125-1133) and the RobotModel mirror passes them
    on every tick: with a MOVING gripper reference (omega != 0) the sim3 switch set still runs on the packed kernel and matches the
    oracle and the general kernel; the other end effectors' references are ignored (their tasks are off)."""
    B = 2048
    models = [wx200, px100]
    cfgs = [common.config("c3", m) for m in models]
    mid = (np.arange(B) % 2).astype(np.int32)
    parts = [common.tick_inputs(m, c, B, seed=91 + i, with_rot=True) for i, (m, c) in enumerate(zip(models, cfgs))]
    d = {k: np.where(mid.reshape((B,) + (1,) * (parts[0][k].ndim - 1)) == 0, parts[0][k], parts[1][k]) for k in parts[0]}
    d["model_id"] = mid
    ref = oracle.tick(models, cfgs, d, DT, B, nthreads=8)
    still = dict(d)
    still["ee_prev_rot"] = d["ee_ref_rot"]                       # resting reference: omega = 0
    ref_still = oracle.tick(models, cfgs, still, DT, B, nthreads=8)
    ok = ref["status"] == 0
    assert np.abs(ref["qdot"] - ref_still["qdot"])[ok].max() > 1e-3   # the moving reference does change the answer
    bt = WbcBatch(models, B)
    for i, c in enumerate(cfgs):
        bt.configure(c, i)
    got = bt.tick(d, DT, want_q_next=True)
    assert bt.stat("last_path") == 2
    bt.set_option("sim3_kernel", 0)
    gen = bt.tick(d, DT, want_q_next=True)
    assert bt.stat("last_path") == 0
    assert (got["status"] == ref["status"]).all() and (gen["status"] == ref["status"]).all()
    e_p, e_g = np.abs(got["qdot"] - ref["qdot"])[ok].max(), np.abs(gen["qdot"] - ref["qdot"])[ok].max()
    print("gripper orientation reference: packed err %.3e, general err %.3e" % (e_p, e_g))
    assert e_p < QDOT_TOL and e_g < QDOT_TOL
    assert np.abs(got["q_next"] - ref["q_next"])[ok].max() < 1e-7
    bt.close()


@pytest.mark.parametrize("cfg_name,with_rot", [("c2", False), ("everything", True), ("c2_three_feet", False)])
def test_orthonormal_contact_presolve_matches_the_oracle(wx200, px100, cfg_name, with_rot):
    """Configurations whose tasks touch the stance legs (BASELINE configs[1], "everything"): the contact equalities are eliminated
    through an orthonormal null-space basis (option presolve_orth, default on) — same status and q̇ as the oracle's
    full-size solve, same as the general path, on both morphologies; `iters` keeps counting the eliminated equalities."""
    B = 1024
    models = [wx200, px100]
    base = "c2" if cfg_name == "c2_three_feet" else cfg_name
    cfgs = [common.config(base, m) for m in models]
    if cfg_name == "c2_three_feet":          # RR in swing: its contact rows and its foot task go, its leg stays a free variable
        for c in cfgs:
            c.con_ee[2] = 0
    mid = (np.arange(B) % 2).astype(np.int32)
    parts = [common.tick_inputs(m, c, B, seed=81 + i, with_rot=with_rot) for i, (m, c) in enumerate(zip(models, cfgs))]
    d = {k: np.where(mid.reshape((B,) + (1,) * (parts[0][k].ndim - 1)) == 0, parts[0][k], parts[1][k]) for k in parts[0]}
    d["model_id"] = mid
    ref = oracle.tick(models, cfgs, d, DT, B, nthreads=8)
    bt = WbcBatch(models, B)
    for i, c in enumerate(cfgs):
        bt.configure(c, i)
    if cfg_name == "c2":
        bt.set_option("packed_orth", 2)      # (1, the default, keeps batches this small on the one-instance kernel: shorter dependent chain)
    got = bt.tick(d, DT, want_q_next=True)
    # config 2 proper runs on the PACKED orth kernel (four instances per wavefront, last_path 3); the others on the general kernel's ORTH variant
    assert bt.stat("last_path") == (3 if cfg_name == "c2" else 0) and bt.stat("last_orth") == 1
    if cfg_name == "c2":
        assert bt.stat("deferred_last") < B // 50                        # (flagged leg blocks: redone by the wave's tail on the general path)
        bt.set_option("packed_orth", 0)
        one = bt.tick(d, DT, want_q_next=True)
        assert bt.stat("last_path") == 0 and bt.stat("last_orth") == 1
        okp = (ref["status"] == 0)
        assert (one["status"] == got["status"]).all() and np.abs(one["qdot"] - got["qdot"])[okp].max() < 1e-6
        assert (one["iters"] == got["iters"]).all()
        bt.set_option("packed_orth", 2)
    bt.set_option("orth_qr", 1)              # the basis through the Householder QR for every instance (by default: flagged leg blocks only)
    qr = bt.tick(d, DT, want_q_next=True)
    bt.set_option("orth_qr", 0)
    bt.set_option("presolve_orth", 0)
    gen = bt.tick(d, DT, want_q_next=True)
    assert bt.stat("last_orth") == 0
    ok = ref["status"] == 0
    assert (qr["status"] == ref["status"]).all()
    e_q = np.abs(qr["qdot"] - ref["qdot"])[ok].max()
    print("%s: basis by QR err %.3e" % (cfg_name, e_q))
    assert e_q < QDOT_TOL and np.abs(qr["qdot"] - got["qdot"])[ok].max() < 1e-6
    assert ok.mean() > 0.9
    assert (got["status"] == ref["status"]).all() and (gen["status"] == ref["status"]).all()
    e_o, e_g = np.abs(got["qdot"] - ref["qdot"])[ok].max(), np.abs(gen["qdot"] - ref["qdot"])[ok].max()
    print("%s: orthonormal presolve err %.3e (iters %.2f), general path err %.3e (iters %.2f), oracle iters %.2f" % (
        cfg_name, e_o, got["iters"][ok].mean(), e_g, gen["iters"][ok].mean(), ref["iters"][ok].mean()))
    assert e_o < QDOT_TOL and e_g < QDOT_TOL
    assert np.abs(got["q_next"] - ref["q_next"])[ok].max() < 1e-7
    assert abs(got["iters"][ok].mean() - ref["iters"][ok].mean()) < 1.01      # (the oracle counts the px100's padded DoF as one more fixed bound)
    # the way out of the presolve (numerically dependent contact rows: never met on valid stances, forced here by a tolerance of 1):
    # every instance falls through to the full-size solve inside the same kernel variant
    bt.set_option("presolve_orth", 1)
    bt.set_option("presolve_tol_exp", 0)
    fb = bt.tick(d, DT, want_q_next=True)
    assert bt.stat("last_orth") == 1
    assert (fb["status"] == gen["status"]).all() and (fb["iters"] == gen["iters"]).all()
    assert np.abs(fb["qdot"] - gen["qdot"]).max() < 1e-12
    bt.close()


def _leg_block_ratio(a):
    """min over the four stance feet of |det K| / (sum |K_ij|)^3 per instance, K = the foot's 3 x 3 leg block of its WORLD-frame
    contact rows (constraint rows 4.. of the sim3 switch set; FR, FL, RR, RL own DoF 9-11, 6-8, 15-17, 12-14)."""
    r = np.full(a["C"].shape[0], np.inf)
    for f, d0 in enumerate((9, 6, 15, 12)):
        K = a["C"][:, 4 + 3 * f:7 + 3 * f, d0:d0 + 3]
        r = np.minimum(r, np.abs(np.linalg.det(K)) / np.abs(K).sum(axis=(1, 2)) ** 3)
    return r


def test_sim3_kernel_pivots_rank_deficient_leg_blocks(wx200):
    """A stance leg whose 3 x 3 WORLD-frame block is (nearly) singular is eliminated with column pivoting by the compact kernel
    (one leg velocity + one contact equality stay in the reduced QP): nothing is deferred. The 64 instances of a 4096-instance
    sample with the smallest |det K| / scale^3 are run at the default threshold (1e-7): whichever fall below it must take that
    path (count checked against the same test on the oracle's C), and every one must match the oracle."""
    B = 4096
    cfg = common.config("c3", wx200)
    d = common.tick_inputs(wx200, cfg, B, seed=77)
    a = oracle.assemble([wx200], [cfg], d, DT, B)
    ratio = _leg_block_ratio(a)
    idx = np.argsort(ratio)[:64]
    sub = {k: v[idx] for k, v in d.items()}
    ref = oracle.tick([wx200], [cfg], sub, DT, len(idx), nthreads=8)
    bt = WbcBatch(wx200, len(idx))
    bt.configure(cfg)
    bt.set_option("count_pivoted", 1)
    bt.set_option("refine", 0)                                              # (the compact kernel is chosen with the refinement off only)
    got = bt.tick(sub, DT)
    assert bt.stat("last_path") == 1 and bt.stat("deferred_last") == 0      # (count_pivoted selects the one-instance compact kernel)
    expect = int((ratio[idx] <= 1e-7).sum())
    assert abs(bt.stat("pivoted_last") - expect) <= 1          # (a ratio within rounding of the threshold may fall either way)
    assert (got["status"] == ref["status"]).all() and (got["status"] >= 0).all()
    ok = ref["status"] == 0
    assert np.abs(got["qdot"] - ref["qdot"])[ok].max() < QDOT_TOL
    print("smallest |det K| / scale^3 in the sample: %.2e, pivoted %d" % (ratio[idx[0]], bt.stat("pivoted_last")))
    bt.close()


@pytest.mark.parametrize("tol_exp", [0, 1, 3, 5])
def test_packed_kernel_pivots_in_place(wx200, px100, tol_exp):
    """The packed kernel (four instances per wavefront) answers a rank-deficient leg block itself: pivoted elimination + SWAP
    of the kept leg velocity into the slot of a base unknown, so the reduced problem keeps its size — at presolve_tol_exp = 0
    every instance swaps on all four legs (four of its six base unknowns become leg velocities), at 3 and 5 the wavefronts mix
    pivoted and plain instances. Nothing may be deferred, and q̇, status, q_next match the oracle."""
    B = 3001
    models = [wx200, px100]
    cfgs = [common.config("c3", m) for m in models]
    mid = (np.arange(B) % 2).astype(np.int32)
    parts = [common.tick_inputs(m, c, B, seed=183 + i) for i, (m, c) in enumerate(zip(models, cfgs))]
    d = {k: np.where(mid.reshape((B,) + (1,) * (parts[0][k].ndim - 1)) == 0, parts[0][k], parts[1][k]) for k in parts[0]}
    d["model_id"] = mid
    ref = oracle.tick(models, cfgs, d, DT, B, nthreads=8)
    bt = WbcBatch(models, B)
    for i, c in enumerate(cfgs):
        bt.configure(c, i)
    bt.set_option("presolve_tol_exp", tol_exp)
    got = bt.tick(d, DT, want_q_next=True)
    assert bt.stat("last_path") == 2 and bt.stat("deferred_last") == 0
    assert (got["status"] == ref["status"]).all()
    ok = ref["status"] == 0
    assert ok.mean() > 0.9
    err = np.abs(got["qdot"] - ref["qdot"])[ok].max()
    print("packed, tol 1e-%d: qdot max-abs err %.3e, working-set changes %.2f (oracle %.2f)" % (tol_exp, err, got["iters"][ok].mean(), ref["iters"][ok].mean()))
    assert err < QDOT_TOL
    assert np.abs(got["q_next"] - ref["q_next"])[ok].max() < 1e-7
    bt.close()


@pytest.mark.parametrize("tol_exp,defer", [(0, 0), (3, 0), (5, 0), (0, 1), (3, 1)])
def test_pivoted_elimination_and_second_pass_match_the_oracle(wx200, px100, tol_exp, defer):
    """Both answers to a rank-deficient stance-leg block under load: option presolve_tol_exp lowers the bar for "rank deficient"
    so that a known share of the batch (all of it, on all four legs, at 0) takes the pivoted elimination inside the compact
    kernel — or, with dbg_force_defer, is handed to the general kernel's second pass over the compact list (the path of a block of
    rank < 2). The count is the one predicted from the oracle's constraint rows; q̇, status and q_next match the oracle."""
    B = 3000                                        # > 2048: the list is longer than the second pass's grid at tol_exp = 0
    models = [wx200, px100]
    cfgs = [common.config("c3", m) for m in models]
    mid = (np.arange(B) % 2).astype(np.int32)
    parts = [common.tick_inputs(m, c, B, seed=83 + i) for i, (m, c) in enumerate(zip(models, cfgs))]
    d = {k: np.where(mid.reshape((B,) + (1,) * (parts[0][k].ndim - 1)) == 0, parts[0][k], parts[1][k]) for k in parts[0]}
    d["model_id"] = mid
    ref = oracle.tick(models, cfgs, d, DT, B, nthreads=8)
    ratio = _leg_block_ratio(oracle.assemble(models, cfgs, d, DT, B))
    expect = int((ratio <= 10.0 ** -tol_exp).sum())
    bt = WbcBatch(models, B)
    for i, c in enumerate(cfgs):
        bt.configure(c, i)
    bt.set_option("presolve_tol_exp", tol_exp)
    bt.set_option("count_pivoted", 1)
    bt.set_option("refine", 0)                      # (the compact kernel and its second pass: chosen with the refinement off only)
    bt.set_option("dbg_force_defer", defer)
    got = bt.tick(d, DT, want_q_next=True)
    n_def, n_piv = bt.stat("deferred_last"), bt.stat("pivoted_last")
    print("tol 1e-%d, force_defer %d: pivoted %d, deferred %d of %d (predicted %d)" % (tol_exp, defer, n_piv, n_def, B, expect))
    n = n_def if defer else n_piv
    assert (n_piv if defer else n_def) == 0
    assert n > 0 and abs(n - expect) <= max(2, expect // 200)
    assert n == B if tol_exp == 0 else n < B
    assert (got["status"] == ref["status"]).all()
    ok = ref["status"] == 0
    assert ok.mean() > 0.9
    err = np.abs(got["qdot"] - ref["qdot"])[ok].max()
    print("   qdot max-abs err %.3e, working-set changes %.2f (oracle %.2f)" % (err, got["iters"][ok].mean(), ref["iters"][ok].mean()))
    assert err < QDOT_TOL
    assert np.abs(got["q_next"] - ref["q_next"])[ok].max() < 1e-7
    bt.close()


def test_refinement_is_what_closes_the_gap_to_the_oracle(wx200, px100):
    """Option `refine` (the analogue of QP_Wrapper.py:37 numRefinementSteps) on every path that carries the step — packed sim3 kernel, general kernel
    with the structural presolve, general kernel at full size (contact rows as equalities), cold and hot-started — on a mixed-morphology batch of the
    benchmark configuration: with it the path agrees with the (refining) oracle to 1e-7, without it to 1e-6 only; status, iteration count and the
    final working set do not depend on it; and the two one-instance kernels, refined, land on the same point as the packed one."""
    B = 1024
    models = [wx200, px100]
    cfgs = [common.config("c3", m) for m in models]
    mid = (np.arange(B) % 2).astype(np.int32)
    parts = [common.tick_inputs(m, c, B, seed=131 + i) for i, (m, c) in enumerate(zip(models, cfgs))]
    d = {k: np.where(mid.reshape((B,) + (1,) * (parts[0][k].ndim - 1)) == 0, parts[0][k], parts[1][k]) for k in parts[0]}
    d["model_id"] = mid
    ref = oracle.tick(models, cfgs, d, DT, B, nthreads=8)
    ok = ref["status"] == 0
    bt = WbcBatch(models, B)
    for i, c in enumerate(cfgs):
        bt.configure(c, i)
    refined = {}
    for name, opts, path in (("packed", {}, 2), ("general + presolve", {"packed_kernel": 0}, 0), ("full size", {"packed_kernel": 0, "presolve": 0}, 0)):
        for k, v in {"packed_kernel": 1, "presolve": 1}.items():
            bt.set_option(k, v)
        for k, v in opts.items():
            bt.set_option(k, v)
        got = {}
        for rf in (1, 0):
            bt.set_option("refine", rf)
            if name == "general + presolve" and rf == 0:
                bt.set_option("sim3_kernel", 0)          # (refine = 0 would bring the compact kernel back: keep the comparison on ONE kernel)
            got[rf] = bt.tick(d, DT, want_working_set=(name != "full size"))
            bt.set_option("sim3_kernel", 1)
            assert bt.stat("last_path") == path, (name, rf, bt.stat("last_path"))
            assert (got[rf]["status"] == ref["status"]).all(), (name, rf)
        e1, e0 = np.abs(got[1]["qdot"] - ref["qdot"])[ok].max(), np.abs(got[0]["qdot"] - ref["qdot"])[ok].max()
        print("%-20s refined %.2e, plain %.2e" % (name, e1, e0))
        assert e1 < REFINED_TOL and e0 < QDOT_TOL and e0 > 5 * e1, (name, e1, e0)
        assert (got[1]["iters"] == got[0]["iters"]).all()
        if name != "full size":
            assert (got[1]["working_set"] == got[0]["working_set"]).all()
            bt.set_option("refine", 1)
            warm = bt.tick(dict(d, working_set=got[1]["working_set"]), DT)          # hot-started with its own set: the WARM variants refine too
            assert (warm["status"] == ref["status"]).all() and np.abs(warm["qdot"] - ref["qdot"])[ok].max() < REFINED_TOL, name
        refined[name] = got[1]["qdot"]
    for name in ("general + presolve", "full size"):
        assert np.abs(refined[name] - refined["packed"])[ok].max() < REFINED_TOL
    bt.close()


def test_packed_orth_ineq_tail_hot_started_matches_cold(wx200):
    """an instance the packed orth kernel's INEQ variant leaves to its tail is redone on the general kernel's ORTH variant when the tick
    is cold and on its warm full-size path when a working set is passed — two numerical routes to one QP. Option orth_qr sends EVERY instance to
    the tail: status and q̇ must agree between the two and with the oracle, and the tail's working set must seed the packed kernel back."""
    B = 300
    cfg = common.config("everything", wx200)
    d = common.tick_inputs(wx200, cfg, B, seed=67, with_rot=True)
    ref = oracle.tick([wx200], [cfg], d, DT, B, nthreads=8)
    ok = ref["status"] == 0
    bt = WbcBatch(wx200, B)
    bt.configure(cfg)
    bt.set_option("packed_orth", 2)
    packed = bt.tick(d, DT, want_working_set=True)                    # the packed path itself (its WARM variant: a set is asked for)
    assert bt.stat("last_path") == 3 and bt.stat("deferred_last") < 0.05 * B
    bt.set_option("orth_qr", 1)
    cold = bt.tick(d, DT)
    assert bt.stat("last_path") == 3 and bt.stat("deferred_last") == B
    warm = bt.tick(dict(d, working_set=packed["working_set"]), DT, want_working_set=True)
    assert bt.stat("last_path") == 3 and bt.stat("deferred_last") == B
    for name, got in (("cold tail", cold), ("hot-started tail", warm), ("packed", packed)):
        assert (got["status"] == ref["status"]).all(), name
        assert np.abs(got["qdot"] - ref["qdot"])[ok].max() < 1e-6, (name, np.abs(got["qdot"] - ref["qdot"])[ok].max())
    assert np.abs(warm["qdot"] - cold["qdot"])[ok].max() < 1e-6
    assert warm["iters"][ok].mean() <= cold["iters"][ok].mean() + 1e-9          # seeded with the optimum's own set: never more working-set changes
    assert (warm["working_set"][ok] == packed["working_set"][ok]).all(axis=1).mean() > 0.97
    bt.set_option("orth_qr", 0)
    back = bt.tick(dict(d, working_set=warm["working_set"]), DT)                # the tail's set seeds the packed kernel
    assert bt.stat("deferred_last") < 0.05 * B and (back["status"] == ref["status"]).all() and np.abs(back["qdot"] - ref["qdot"])[ok].max() < 1e-6
    bt.close()


@pytest.mark.parametrize("tol_exp,warm", [(0, 0), (3, 0), (3, 1)])
def test_packed_kernel_redoes_what_it_cannot_reduce_in_its_own_tail(wx200, px100, tol_exp, warm):
    """The packed kernel launches no second pass: an instance it leaves out (a stance-leg block of rank < 2; here every flagged block, through
    dbg_force_defer, with the bar for "flagged" lowered by presolve_tol_exp) is redone by its own wave on the general path at the end of the
    SAME kernel. Count as predicted from the oracle's constraint rows, answers the oracle's — cold and with working sets in and out."""
    B = 3001
    models = [wx200, px100]
    cfgs = [common.config("c3", m) for m in models]
    mid = (np.arange(B) % 2).astype(np.int32)
    parts = [common.tick_inputs(m, c, B, seed=83 + i) for i, (m, c) in enumerate(zip(models, cfgs))]
    d = {k: np.where(mid.reshape((B,) + (1,) * (parts[0][k].ndim - 1)) == 0, parts[0][k], parts[1][k]) for k in parts[0]}
    d["model_id"] = mid
    ref = oracle.tick(models, cfgs, d, DT, B, nthreads=8)
    ratio = _leg_block_ratio(oracle.assemble(models, cfgs, d, DT, B))
    expect = int((ratio <= 10.0 ** -tol_exp).sum())
    bt = WbcBatch(models, B)
    for i, c in enumerate(cfgs):
        bt.configure(c, i)
    bt.set_option("presolve_tol_exp", tol_exp)
    bt.set_option("dbg_force_defer", 1)
    got = bt.tick(d, DT, want_q_next=True, want_working_set=bool(warm))
    assert bt.stat("last_path") == 2
    n = bt.stat("deferred_last")
    print("packed, tol 1e-%d, warm %d: %d of %d instances redone in the tail (predicted %d)" % (tol_exp, warm, n, B, expect))
    assert n > 0 and abs(n - expect) <= max(2, expect // 200) and (n == B if tol_exp == 0 else n < B)
    ok = ref["status"] == 0
    assert (got["status"] == ref["status"]).all() and ok.mean() > 0.9
    assert np.abs(got["qdot"] - ref["qdot"])[ok].max() < QDOT_TOL and np.abs(got["q_next"] - ref["q_next"])[ok].max() < 1e-7
    if warm:      # the tail's working sets are the packed rows' (full-problem indexing on every path): seeding the next tick with them works
        again = bt.tick(dict(d, working_set=got["working_set"]), DT, want_working_set=True)
        assert (again["status"] == ref["status"]).all() and np.abs(again["qdot"] - ref["qdot"])[ok].max() < QDOT_TOL
        assert (again["working_set"][ok] == got["working_set"][ok]).all(axis=1).mean() > 0.98
    bt.set_option("dbg_force_defer", 0)
    bt.tick(d, DT)
    assert bt.stat("last_path") == 2 and bt.stat("deferred_last") == 0      # (the statistic is per launch: nothing carried over)
    bt.close()


@pytest.mark.parametrize("cfg_name", ["c2", "everything", "c3"])
def test_tick_and_assemble_on_the_matrix_cores(wx200, cfg_name):
    """Option jtj_mfma: H = A'A of wbc_assemble / wbc_tick from v_mfma_f64_16x16x4_f64 (QP_Wrapper.py:17) against the oracle,
    and the option really selects the kernel that has the matrix-core path (the compact sim3 kernel has none)."""
    B = 512
    cfg = common.config(cfg_name, wx200)
    d = common.tick_inputs(wx200, cfg, B, seed=19, with_rot=(cfg_name == "everything"))
    ref = oracle.tick([wx200], [cfg], d, DT, B, nthreads=8)
    ar = oracle.assemble([wx200], [cfg], d, DT, B)
    bt = WbcBatch(wx200, B)
    bt.configure(cfg)
    valu = bt.tick(d, DT)
    path_default = bt.stat("last_path")
    bt.set_option("jtj_mfma", 1)
    a = bt.assemble(d, DT)
    for k in ("A", "b", "H", "g", "C", "Clb", "Cub", "lb", "ub"):
        assert relerr(a[k], ar[k]) < 1e-11, (k, relerr(a[k], ar[k]))
    got = bt.tick(d, DT, want_q_next=True)
    assert bt.stat("last_path") == 0                                  # general kernel
    assert path_default == (2 if cfg_name == "c3" else 0)               # 2: the packed compact kernel (four instances per wavefront)
    assert (got["status"] == ref["status"]).all()
    ok = ref["status"] == 0
    assert np.abs(got["qdot"] - ref["qdot"])[ok].max() < QDOT_TOL
    assert np.abs(got["qdot"] - valu["qdot"])[ok].max() < QDOT_TOL
    assert np.abs(got["q_next"] - ref["q_next"])[ok].max() < 1e-7
    bt.close()


@pytest.mark.parametrize("cfg_name", ["c3", "everything"])
def test_corrected_damper_map_on_the_device(wx200, px100, cfg_name):
    """velDamperJointConstraints with the intended index map (damper_compat=False; the reference's own map is off by one,
    SURVEY.md C.3): bounds and the tick on the HIP path against the oracle, both morphologies."""
    B = 512
    models = [wx200, px100]
    if cfg_name == "c3":
        cfgs = [wbc_model.sim3_config(m, damper_compat=False) for m in models]
    else:
        cfgs = [wbc_model.make_config(m, Trunk=True, FR=True, FL=True, RR=True, RL=True, Grip=True, Joint="PREV", task_com=True,
                                      cCoM=True, cTrunk=True, cFR=True, cFL=True, cRR=True, cRL=True, mode="static_reach",
                                      damper_compat=False) for m in models]
    assert list(cfgs[0].damper_qidx[:8]) == [0, 1, 2, 3, 4, 5, 7, 8]
    for i, (m, c) in enumerate(zip(models, cfgs)):
        d = common.tick_inputs(m, c, B, seed=43 + i)
        compat = common.config("c3", m) if cfg_name == "c3" else common.config("everything", m)
        ref, ar = oracle.tick([m], [c], d, DT, B, nthreads=8), oracle.assemble([m], [c], d, DT, B)
        bt = WbcBatch(m, B)
        bt.configure(c)
        a = bt.assemble(d, DT)
        for k in ("lb", "ub", "C", "Clb", "Cub", "H", "g"):
            assert relerr(a[k], ar[k]) < 1e-11, (k, relerr(a[k], ar[k]))
        assert np.abs(ar["lb"] - oracle.assemble([m], [compat], d, DT, B)["lb"]).max() > 1e-3    # the two maps do differ on this batch
        got = bt.tick(d, DT, want_q_next=True)
        assert (got["status"] == ref["status"]).all()
        ok = ref["status"] == 0
        assert ok.mean() > 0.9 and np.abs(got["qdot"] - ref["qdot"])[ok].max() < QDOT_TOL
        bt.close()


def test_fk_outputs_of_a_mixed_batch_with_the_small_model_first(wx200, px100):
    """FK output strides are the LARGEST model's joint / frame counts whatever the order of the models in the handle:
    with [px100 (21 joints), wx200 (22)] every instance's oMi / oMf rows must be its own (round 1 used model 0's count as
    the stride, so a wx200 instance wrote its 22nd joint into its neighbour's first row)."""
    import wbc_workload
    B = 257
    models = [px100, wx200]
    rng = np.random.default_rng(6)
    mid = (np.arange(B) % 2).astype(np.int32)
    qs = [wbc_workload.sample_q(m, B, rng) for m in models]
    q = np.where(mid[:, None] == 0, qs[0], qs[1])
    ref = oracle.fk(models, q, mid)
    bt = WbcBatch(models, B)
    got = bt.fk(q, mid)
    assert got["oMi"].shape == (B, 22, 12) and ref["oMi"].shape == (B, 22, 12)
    for k in ("oMi", "oMf", "J", "com", "Jcom"):
        assert np.abs(got[k] - ref[k]).max() < 1e-12, k
    assert (got["oMi"][mid == 0, 21] == 0).all() and np.abs(got["oMi"][mid == 1, 21]).max() > 0
    bt.close()


def test_wrong_shapes_and_devices_are_refused_before_any_launch(wx200, px100):
    """The kernels index raw pointers with fixed per-instance strides: every array's shape is checked at the Python boundary."""
    import torch
    B = 16
    cfg = common.config("c3", wx200)
    d = common.tick_inputs(wx200, cfg, B, seed=2)
    bt = WbcBatch(wx200, 64)
    bt.configure(cfg)
    with pytest.raises(capi.WbcError, match=r"q must be \[B, 27\]"):
        bt.tick(dict(d, q=d["q"][:, :26]), DT)                          # px100's true nq instead of the padded 27
    with pytest.raises(capi.WbcError, match="ee_target"):
        bt.tick(dict(d, ee_target=d["ee_target"][:, :4]), DT)
    with pytest.raises(capi.WbcError, match="trunk_box_center"):
        bt.tick(dict(d, trunk_box_center=d["trunk_box_center"][: B - 1]), DT)
    with pytest.raises(capi.WbcError, match="model_id"):
        bt.tick(dict(d, model_id=np.zeros(B - 3, dtype=np.int32)), DT)
    with pytest.raises(capi.WbcError, match="qdot"):
        bt.tick(d, DT, out=dict(qdot=np.zeros((B, 25)), status=np.zeros(B, np.int32)))
    with pytest.raises(capi.WbcError, match="unknown tick inputs"):
        bt.tick(dict(d, trunk_taget=d["trunk_target"]), DT)
    with pytest.raises(capi.WbcError, match="ee_target_step"):
        bt.rollout(d, DT, 2, ee_target_step=np.zeros((B, 4, 3)))
    with pytest.raises(capi.WbcError, match="C must be"):
        bt.qp_solve(np.eye(5)[None], np.zeros((1, 5)), np.zeros((1, 2, 4)), None, None, np.zeros((1, 2)), np.zeros((1, 2)))
    with pytest.raises(capi.WbcError, match="mixing host and device"):
        bt.tick(dict(d, q=torch.from_numpy(d["q"]).cuda()), DT)
    assert bt.tick(d, DT)["status"].shape == (B,)                      # and the well-formed call still works
    bt.close()


def test_tick_mixed_morphology(wx200, px100):
    """BASELINE config 5: wx200 (nv 26) and px100 (nv 25, padded DoF) interleaved lane by lane."""
    B = 1024
    models = [wx200, px100]
    cfgs = [common.config("c3", m) for m in models]
    mid = (np.arange(B) % 2).astype(np.int32)
    parts = [common.tick_inputs(m, c, B, seed=31 + i) for i, (m, c) in enumerate(zip(models, cfgs))]
    d = {k: np.where(mid.reshape((B,) + (1,) * (parts[0][k].ndim - 1)) == 0, parts[0][k], parts[1][k]) for k in parts[0]}
    d["model_id"] = mid
    ref = oracle.tick(models, cfgs, d, DT, B, nthreads=8)
    bt = WbcBatch(models, B)
    for i, c in enumerate(cfgs):
        bt.configure(c, i)
    got = bt.tick(d, DT, want_q_next=True)
    assert (got["status"] == ref["status"]).all()
    ok = ref["status"] == 0
    assert np.abs(got["qdot"] - ref["qdot"])[ok].max() < QDOT_TOL
    assert (got["qdot"][mid == 1, 25] == 0).all()
    assert np.abs(got["q_next"] - ref["q_next"])[ok].max() < 1e-7
    bt.close()


def test_integrate_parity(wx200):
    rng = np.random.default_rng(9)
    import wbc_workload
    B = 256
    q = wbc_workload.sample_q(wx200, B, rng)
    v = rng.normal(size=(B, 26)) * 2
    v[:8, 3:6] = 0                       # exercise the small-angle branch
    ref = oracle.integrate([wx200], q, v, DT)
    bt = WbcBatch(wx200, B)
    got = bt.integrate(q, v, DT)
    assert np.abs(got - ref).max() < 1e-13
    big = rng.normal(size=(B, 26)) * 300  # large rotations per step: exercises every quaternion branch
    assert np.abs(bt.integrate(q, big, DT) - oracle.integrate([wx200], q, big, DT)).max() < 1e-12
    bt.close()


@pytest.mark.parametrize("B", [512, 5, 63])
def test_update_state_parity(wx200, px100, B):
    """The tail of runWBC: updateState(running=True) + trunkWorldPos (Robot_Wrapper4.py:387-428, 1297-1327); batch sizes that do
    not fill the last wavefront of the four-instances-per-wave kernel included."""
    import wbc_workload
    rng = np.random.default_rng(13)
    models = [wx200, px100]
    mid = (np.arange(B) % 2).astype(np.int32)
    qa = [wbc_workload.sample_q(m, B, rng) for m in models]
    qb = [wbc_workload.sample_q(m, B, rng) for m in models]
    q_cur = np.where(mid[:, None] == 0, qa[0], qa[1])
    q_next = np.where(mid[:, None] == 0, qb[0], qb[1])
    imu = rng.normal(size=(B, 4))
    imu /= np.linalg.norm(imu, axis=1, keepdims=True)
    targets = rng.normal(size=(B, 5, 3))
    bt = WbcBatch(models, B)
    for i, m in enumerate(models):
        bt.configure(common.config("c3", m), i)
    for packed in (1, 0):                    # four instances per wavefront (the sim3 family's plans allow it) / one per wavefront
        bt.set_option("packed_update", packed)
        for im in (imu, None):
            ref = oracle.update_state(models, q_cur, q_next, targets, im, mid)
            got = bt.update_state(q_cur, q_next, targets, im, mid)
            assert bt.stat("last_update_packed") == packed
            assert np.abs(got - ref).max() < 1e-13
            assert (got[:, 3:] == ref[:, 3:]).all()
    bt.close()


@pytest.mark.parametrize("cfg_name,K,with_imu", [("c3", 12, True), ("c3", 5, False), ("c3_hybrid", 6, True), ("everything", 6, True), ("c3_trunk_task", 8, True),
                                                  ("c3_mani", 4, True), ("c2", 5, True), ("full", 5, True)])
def test_rollout_parity(wx200, cfg_name, K, with_imu):
    """K closed-loop ticks on the device against the oracle's tick / update_state / state-advance loop:
    state, targets, worst status, iteration total and the gripper trace."""
    B = 192
    cfg = common.config(cfg_name, wx200)
    d = common.tick_inputs(wx200, cfg, B, seed=37, with_rot=(cfg_name in ("everything", "c3_trunk_task", "full")))
    rng = np.random.default_rng(2)
    step = np.zeros((B, 5, 3))
    step[:, 4] = rng.normal(0, 1e-4, (B, 3))
    tstep = rng.normal(0, 5e-5, (B, 3))
    imu = d["q"][:, 3:7].copy() if with_imu else None
    ref = oracle.rollout([wx200], [cfg], d, DT, B, K, ee_target_step=step, trunk_target_step=tstep, imu=imu, nthreads=8)
    bt = WbcBatch(wx200, B)
    bt.configure(cfg)
    before = {k: v.copy() for k, v in d.items()}
    ok = ref["status"] == 0
    assert ok.mean() > 0.8
    res = {}
    for warm in (1, 0):      # 1: every tick seeded with the previous tick's working set (f2; off by default); 0: cold, like the oracle's ticks
        bt.set_option("warm_start", warm)
        got = res[warm] = bt.rollout(d, DT, K, ee_target_step=step, trunk_target_step=tstep, imu=imu)
        if cfg_name in ("c3", "c3_hybrid", "c3_trunk_task", "c3_mani"):
            assert bt.stat("last_path") == 2, warm                  # warm or cold, the roll-out stays on the packed kernel (TRUNK / QCON variants included)
            assert bt.stat("last_update_packed") == 1               # ... and so does its state update (trunk reference state included)
        if cfg_name == "c2":
            assert bt.stat("last_update_packed") == 1               # (configs[1]: the state update is packed whatever kernel the tick ran on)
        if cfg_name == "full":                                      # the warm-up problem: packed box kernel (its WARM variant when hot-started) + packed update
            assert bt.stat("last_path") == 4 and bt.stat("last_update_packed") == 1, warm
        assert all((d[k] == before[k]).all() for k in d)            # in0 is only read
        assert (got["status"] == ref["status"]).all(), warm
        # one tick agrees to ~1e-6 in qdot (cond(H) ~ 3e9); K ticks of dt = 2 ms integrate that into ~1e-8 of state
        assert np.abs(got["q"] - ref["q"])[ok].max() < 1e-6, warm
        assert np.abs(got["qdot"] - ref["qdot"])[ok].max() < 10 * QDOT_TOL, warm
        assert np.abs(got["ee_target"] - ref["ee_target"]).max() < 1e-15
        assert np.abs(got["grip_trace"] - ref["grip_trace"])[:, ok].max() < 1e-6, warm
        if imu is not None:
            assert (got["q"][:, 3:7] == imu).all()
    assert (res[0]["iters"][ok] - ref["iters"][ok]).__abs__().max() <= 2 * K       # cold: the oracle's working-set changes
    print("%s: working-set changes per tick: cold %.2f, warm-started %.2f (oracle %.2f)" % (
        cfg_name, res[0]["iters"][ok].mean() / K, res[1]["iters"][ok].mean() / K, ref["iters"][ok].mean() / K))
    bt.close()


def test_rollout_equals_tick_plus_update_state(wx200):
    """wbc_rollout is nothing but the entry points chained: replay it through wbc_tick / wbc_update_state from the host."""
    B, K = 64, 4
    cfg = common.config("c3", wx200)
    d = common.tick_inputs(wx200, cfg, B, seed=39)
    step = np.zeros((B, 5, 3))
    step[:, 4] = [1e-4, -5e-5, 2e-5]
    bt = WbcBatch(wx200, B)
    bt.configure(cfg)
    bt.set_option("warm_start", 1)
    got = bt.rollout(d, DT, K, ee_target_step=step)
    s = {k: v.copy() for k, v in d.items()}
    for _ in range(K):
        o = bt.tick(s, DT, want_q_next=True, want_working_set=True)
        s["q"] = bt.update_state(s["q"], o["q_next"], s["ee_target"])
        s["prev_ee_target"][:, 4] = s["ee_target"][:, 4]
        s["ee_target"] = s["ee_target"] + step
        s["working_set"] = o["working_set"]               # the roll-out hot-starts every tick from the previous one's working set
    assert (got["q"] == s["q"]).all() and (got["qdot"] == o["qdot"]).all() and (got["ee_target"] == s["ee_target"]).all()
    bt.set_option("warm_start", 0)                         # ... unless told not to: then it is the chain of cold ticks
    cold = bt.rollout(d, DT, K, ee_target_step=step)
    s = {k: v.copy() for k, v in d.items()}
    for _ in range(K):
        o = bt.tick(s, DT, want_q_next=True)
        s["q"] = bt.update_state(s["q"], o["q_next"], s["ee_target"])
        s["prev_ee_target"][:, 4] = s["ee_target"][:, 4]
        s["ee_target"] = s["ee_target"] + step
    assert (cold["q"] == s["q"]).all() and (cold["qdot"] == o["qdot"]).all()
    assert np.abs(cold["q"] - got["q"]).max() < 1e-7       # same minimiser either way
    bt.close()


@pytest.mark.parametrize("cfg_name", ["c3", "c2", "full"])
def test_full_size_properties(wx200, cfg_name):
    """BASELINE's full single-GPU size (B = 65536; config 3 on the packed kernel, config 2's switch set on the packed orth kernel, the warm-up
    problem on the packed box kernel): size-independent certificates on every instance (contact rows satisfied, bounds and box rows
    respected; the warm-up problem: the KKT conditions of its bound-constrained QP from the device's own H and g) + oracle parity on a random subsample."""
    B = 65536
    cfg = common.config(cfg_name, wx200)
    d = common.tick_inputs(wx200, cfg, B, seed=41, with_rot=(cfg_name == "full"))
    bt = WbcBatch(wx200, B)
    bt.configure(cfg)
    got = bt.tick(d, DT)
    a = bt.assemble(d, DT, want=("C", "Clb", "Cub", "lb", "ub"))
    ok = got["status"] == 0
    assert ok.mean() > 0.98
    x = got["qdot"]
    Cx = np.einsum("bpn,bn->bp", a["C"], x)
    scale = 1 + np.abs(x).max(axis=1, keepdims=True)
    c0 = 4 if cfg_name == "c3" else 0                                            # (config 2 has no trunk box in front of the contact rows)
    if cfg_name != "full":
        assert (np.abs(Cx[:, c0:])[ok] / scale[ok]).max() < 1e-8                 # 12 contact equalities
    assert bt.stat("last_path") == {"c3": 2, "c2": 3, "full": 4}[cfg_name] and bt.stat("last_orth") == (1 if cfg_name == "c2" else 0)
    if cfg_name == "full":
        # stationarity of the bound-constrained QP on every instance: r = H x + g vanishes on the free DoF, points inwards on the held ones
        assert a["C"].shape[1] == 0 and bt.stat("deferred_last") == 0
        hg = bt.assemble(d, DT, want=("H", "g"))
        r = np.einsum("bij,bj->bi", hg["H"], x) + hg["g"]
        at_lb = np.abs(x - a["lb"]) <= 1e-9 * np.maximum(1, np.abs(a["lb"]))
        at_ub = np.abs(x - a["ub"]) <= 1e-9 * np.maximum(1, np.abs(a["ub"]))
        rs = np.abs(hg["H"]).max(axis=(1, 2))[:, None] * scale
        assert (np.abs(r)[~(at_lb | at_ub) & ok[:, None]] / np.broadcast_to(rs, r.shape)[~(at_lb | at_ub) & ok[:, None]]).max() < 1e-9
        assert ((-r / rs)[at_lb & ~at_ub & ok[:, None]]).max() < 1e-9 and ((r / rs)[at_ub & ~at_lb & ok[:, None]]).max() < 1e-9
    if a["C"].shape[1]:
        assert ((a["Clb"] - Cx)[ok] / scale[ok]).max() < 1e-8 and ((Cx - a["Cub"])[ok] / scale[ok]).max() < 1e-8
    assert ((a["lb"] - x)[ok]).max() < 1e-8 and ((x - a["ub"])[ok]).max() < 1e-8
    rng = np.random.default_rng(0)
    idx = rng.choice(B, 1024, replace=False)
    sub = {k: v[idx] for k, v in d.items()}
    ref = oracle.tick([wx200], [cfg], sub, DT, len(idx), nthreads=8)
    assert (ref["status"] == got["status"][idx]).all()
    good = ref["status"] == 0
    assert np.abs(ref["qdot"] - x[idx])[good].max() < QDOT_TOL
    bt.close()


def test_api_misuse_is_refused_with_a_message(wx200, px100):
    """Return-code contract of include/wbc.h: misuse gives a negative code + wbc_last_error(), never a launch."""
    B = 8
    cfg = common.config("c3", wx200)
    d = common.tick_inputs(wx200, cfg, B, seed=1)
    bt = WbcBatch(wx200, B)
    with pytest.raises(capi.WbcError, match="no configuration"):
        bt.tick(d, DT)                                               # tick before wbc_batch_configure
    bt.configure(cfg)
    with pytest.raises(capi.WbcError, match="dt > 0"):
        bt.tick(d, 0.0)
    big = common.tick_inputs(wx200, cfg, B + 1, seed=1)
    with pytest.raises(capi.WbcError, match="max_batch"):
        bt.tick(big, DT)                                             # B beyond the handle's workspace
    bad = {k: v for k, v in d.items() if k != "trunk_box_center"}
    with pytest.raises(capi.WbcError, match="trunk_box_center"):
        bt.tick(bad, DT)                                             # trunk constraint without its box centre
    bad = {k: v for k, v in d.items() if k != "prev_ee_target"}
    with pytest.raises(capi.WbcError, match="prev_ee_target"):
        bt.tick(bad, DT)
    off = wbc_model.sim3_config(wx200, Joint=False)
    with pytest.raises(capi.WbcError, match="posture task must be on"):
        bt.configure(off)                                            # H = J'J alone is singular (RW4:1199-1206)
    with pytest.raises(capi.WbcError, match="unknown option"):
        bt.set_option("no_such_knob", 1)
    with pytest.raises(capi.WbcError, match="ticks >= 1"):
        bt.rollout(d, DT, 0)
    H = np.eye(27)[None]
    with pytest.raises(capi.WbcError, match="out of range"):
        bt.qp_solve(H, np.zeros((1, 27)))                            # n > 26
    # two morphologies must share the switches, and a mixed batch needs model_id
    b2 = WbcBatch([wx200, px100], B)
    b2.configure(common.config("c3", wx200), 0)
    b2.configure(common.config("c2", px100), 1)                     # accepted: a batch may change switch sets model by model ...
    with pytest.raises(capi.WbcError, match="share the task/constraint switches"):
        b2.tick(dict(d, model_id=np.zeros(B, dtype=np.int32)), DT)    # ... but it cannot be USED while the models disagree
    # equal row counts are not enough (round 1 compared only m and p): CoM box + trunk box vs two contact feet, both p = 6
    six_a = wbc_model.make_config(wx200, Grip=True, Joint="PREV", cCoM=True, cTrunk=True, mode="static_reach")
    six_b = wbc_model.make_config(px100, Grip=True, Joint="PREV", cFR=True, cFL=True, mode="static_reach")
    b2.configure(six_a, 0)
    b2.configure(six_b, 1)
    with pytest.raises(capi.WbcError, match="share the task/constraint switches"):
        b2.tick(dict(d, model_id=np.zeros(B, dtype=np.int32)), DT)
    b2.configure(common.config("c3", wx200), 0)
    b2.configure(common.config("c3", px100), 1)
    with pytest.raises(capi.WbcError, match="model_id is required"):
        b2.tick(d, DT)
    # a status of its own is never visible: WBC_QP_DEFERRED is internal to the two-pass tick
    out = bt.tick(d, DT)
    assert out["status"].min() >= 0
    bt.close()
    b2.close()


def test_qp_packed_edge_cases_agree_with_the_oracle_and_the_one_per_wavefront_kernel():
    """The packed stand-alone QP kernel (csrc/wbc_k_qpp.hip) on the problems a plug-in boundary has to survive, mixed in one batch so that the two or four
    problems of a wavefront differ in fate: infeasible rows, contradictory and redundant equality rows, more equality rows than unknowns, H not positive
    definite, a NaN bound, no inequality active at all — and the extreme shapes (n = 1; n = 26 with p = 24 rows and m = 96 task rows; no box, no rows).
    Status from the oracle, x = 0 where unsolved, the neighbours' answers untouched; the one-per-wavefront kernel says the same."""
    rng = np.random.default_rng(314)
    for n, p, m in ((14, 9, 20), (26, 24, 96), (1, 0, 1), (5, 8, 7)):
        B = 67
        A = rng.normal(size=(B, m, n))
        b = rng.normal(size=(B, m))
        C = rng.normal(size=(B, p, n)) if p else None
        lb, ub = -rng.uniform(0.3, 1.0, (B, n)), rng.uniform(0.3, 1.0, (B, n))
        cl, cu = (-rng.uniform(0.3, 1.0, (B, p)), rng.uniform(0.3, 1.0, (B, p))) if p else (None, None)
        if m < n:
            A = np.concatenate([A, np.broadcast_to(0.1 * np.eye(n), (B, n, n))], axis=1)
            b = np.concatenate([b, np.zeros((B, n))], axis=1)
        kinds = np.arange(B) % 8
        if p:
            cl[kinds == 1, 0], cu[kinds == 1, 0] = 50.0, 60.0                                  # infeasible against the box
            if p >= 2:
                C[kinds == 2, 1] = C[kinds == 2, 0]                                            # contradictory equalities
                cl[kinds == 2, 0] = cu[kinds == 2, 0] = 0.1
                cl[kinds == 2, 1] = cu[kinds == 2, 1] = -0.1
                C[kinds == 3, 1] = 2.0 * C[kinds == 3, 0]                                      # a redundant equality (consistent)
                cl[kinds == 3, 0] = cu[kinds == 3, 0] = 0.05
                cl[kinds == 3, 1] = cu[kinds == 3, 1] = 0.1
            if p > n:
                cl[kinds == 4] = cu[kinds == 4] = 0.0                                          # more equality rows than unknowns, all through the origin: consistent
        A[kinds == 5] = 0.0                                                                    # H = 0: not positive definite
        lb[kinds == 6, 0] = np.nan
        lb[kinds == 7], ub[kinds == 7] = -1e3, 1e3                                             # nothing active but (perhaps) rows
        xr, sr, ir = oracle.qp_solve_ls(A, b, C, lb, ub, cl, cu)
        bt = WbcBatch([], B)
        x, st, it = bt.qp_solve_ls(A, b, C, lb, ub, cl, cu)
        assert bt.stat("last_qp_path") == (4 if (n <= 16 and p <= 16) else 2)
        assert (st == sr).all(), (n, p, np.nonzero(st != sr), st[st != sr], sr[st != sr])
        assert len(set(sr.tolist())) >= 2 and (x[sr != 0] == 0).all()
        ok = sr == 0
        assert np.abs(x - xr)[ok].max() < 1e-8 and (it[ok] == ir[ok]).all()
        bt.set_option("packed_kernel", 0)
        x1, st1, it1 = bt.qp_solve_ls(A, b, C, lb, ub, cl, cu)
        assert (st1 == sr).all() and np.abs(x1 - x)[ok].max() < 1e-8
        # no box and no rows at all: the unconstrained minimiser
        bt.set_option("packed_kernel", 1)
        x2, st2, it2 = bt.qp_solve_ls(A[kinds != 5], b[kinds != 5])
        assert (st2 == 0).all() and (it2 == 0).all()
        xs = np.stack([np.linalg.lstsq(A_, b_, rcond=None)[0] for A_, b_ in zip(A[kinds != 5], b[kinds != 5])])
        assert np.abs(x2 - xs).max() < 1e-9
        bt.close()


def test_infeasible_and_degenerate_instances_agree_with_the_oracle(wx200):
    """Instances the QP cannot satisfy (trunk far outside its box: the box rows contradict the bounds) must be flagged
    exactly like the oracle flags them, and must not disturb their neighbours in the batch."""
    B = 256
    cfg = common.config("c3", wx200)
    d = common.tick_inputs(wx200, cfg, B, seed=61)
    d["trunk_box_center"] = d["trunk_box_center"].copy()
    d["trunk_box_center"][::7, 0] *= 3.0            # box centre z three times the trunk height: |rate| >> velocity bounds
    d["trunk_box_center"][3::11, 1] += 1.0          # roll box one radian away
    ref = oracle.tick([wx200], [cfg], d, DT, B, nthreads=8)
    assert (ref["status"] == capi.QP_INFEASIBLE).sum() > 10 and (ref["status"] == 0).sum() > 100
    bt = WbcBatch(wx200, B)
    bt.configure(cfg)
    for sim3 in (1, 0):
        bt.set_option("sim3_kernel", sim3)
        got = bt.tick(d, DT)
        assert (got["status"] == ref["status"]).all(), sim3
        ok = ref["status"] == 0
        assert np.abs(got["qdot"] - ref["qdot"])[ok].max() < QDOT_TOL
        # an unsolved QP returns q̇ = 0 (what the reference's xOpt holds on its first QP, QP_Wrapper.py:50-51): hold still
        assert (got["qdot"][~ok] == 0).all() and (ref["qdot"][~ok] == 0).all()
    bt.close()


@pytest.mark.parametrize("posture", ["TIKHONOV", "PREV"])
def test_packed_orth_kernel_variants_and_non_finite_inputs(wx200, px100, posture):
    """wbc_tick_orthp_kernel beyond BASELINE configs[1]'s own switch set: the PREV posture target (its part of g goes through Z), a mixed
    wx200 / px100 batch at a size the default policy sends to it (B >= 4608), ragged tail (B not a multiple of four), q_next, and non-finite
    inputs — an instance with a NaN / Inf in its state is flagged like the oracle flags it, returns q̇ = 0 and does not disturb the three
    other instances of its wavefront."""
    B = 4610
    models = [wx200, px100]
    cfgs = [wbc_model.make_config(m, FR=True, FL=True, RR=True, RL=True, Grip=True, Joint=(True if posture == "TIKHONOV" else "PREV"), task_com=True,
                                  cFR=True, cFL=True, cRR=True, cRL=True, use_bounds=False) for m in models]
    mid = (np.arange(B) % 2).astype(np.int32)
    # (PREV: also with MOVING orientation references for the five EE tasks — calcTargetVelEE3's omega feed-forward, RW4:1125-1133)
    parts = [common.tick_inputs(m, c, B, seed=31 + i, with_rot=(posture == "PREV")) for i, (m, c) in enumerate(zip(models, cfgs))]
    d = {k: np.where(mid.reshape((B,) + (1,) * (parts[0][k].ndim - 1)) == 0, parts[0][k], parts[1][k]) for k in parts[0]}
    d["model_id"] = mid
    d["q"] = d["q"].copy()
    d["q"][5, 9] = np.nan                       # a leg angle
    d["q"][1002, 4] = np.inf                    # a quaternion component
    d["ee_target"] = d["ee_target"].copy()
    d["ee_target"][2007, 4, 1] = np.nan         # the gripper target
    ref = oracle.tick(models, cfgs, d, DT, B, nthreads=8)
    bad = np.zeros(B, bool)
    bad[[5, 1002, 2007]] = True
    assert (ref["status"][bad] != 0).all() and (ref["status"][~bad] == 0).mean() > 0.99
    bt = WbcBatch(models, B)
    for i, c in enumerate(cfgs):
        bt.configure(c, i)
    got = bt.tick(d, DT, want_q_next=True)
    assert bt.stat("last_path") == 3 and bt.stat("last_orth") == 1     # the default policy: packed from 4608 instances on
    ok = ref["status"] == 0
    assert ((got["status"] != 0) == (ref["status"] != 0)).all()
    assert (got["qdot"][~ok] == 0).all() and np.isfinite(got["qdot"]).all()
    err = np.abs(got["qdot"] - ref["qdot"])[ok].max()
    print("packed orth kernel, posture %s, mixed batch: qdot max-abs err %.3e" % (posture, err))
    assert err < QDOT_TOL and np.abs(got["q_next"] - ref["q_next"])[ok].max() < 1e-7
    assert (got["iters"][ok] == 12).all()
    bt.set_option("packed_orth", 0)
    one = bt.tick(d, DT)
    assert bt.stat("last_path") == 0 and np.abs(one["qdot"] - got["qdot"])[ok].max() < 1e-6
    bt.close()


@pytest.mark.parametrize("variant", ["everything", "no_com_box", "no_trunk"])
def test_packed_orth_kernel_with_inequality_rows(wx200, px100, variant):
    """wbc_tick_orthp_kernel<INEQ>: the task problems whose tasks touch the stance legs AND that keep inequality rows — trunk box, CoM
    box, the velocity box of every DoF (in the reduced coordinates: rows of Z, two per lane), with the trunk task — four instances per wavefront:
    tests/common.py "everything" and relatives (no CoM box; no trunk task / trunk box; with three stance feet the reduced problem has 14 unknowns, more than
    the variant's 12: general kernel) on a mixed wx200 / px100 batch at a size the default policy sends to it, ragged tail, q_next, the oracle's working-set changes, infeasible
    instances (a trunk outside its box) reported as the oracle reports them, and non-finite inputs contained to their own instance."""
    B = 4610
    models = [wx200, px100]
    kw = dict(everything=dict(Trunk=True, FR=True, FL=True, RR=True, RL=True, Grip=True, Joint="PREV", task_com=True, cCoM=True, cTrunk=True,
                              cFR=True, cFL=True, cRR=True, cRL=True, mode="static_reach"),
              no_com_box=dict(Trunk=True, FR=True, FL=True, RR=True, RL=True, Grip=True, Joint="PREV", task_com=True, cTrunk=True,
                              cFR=True, cFL=True, cRR=True, cRL=True, mode="static_reach"),
              no_trunk=dict(FR=True, FL=True, RR=True, RL=True, Grip=True, Joint=True, task_com=True, cCoM=True,
                            cFR=True, cFL=True, cRR=True, cRL=True, mode="static_reach"))[variant]
    cfgs = [wbc_model.make_config(m, **kw) for m in models]
    mid = (np.arange(B) % 2).astype(np.int32)
    parts = [common.tick_inputs(m, c, B, seed=61 + i, with_rot=True) for i, (m, c) in enumerate(zip(models, cfgs))]
    d = {k: np.where(mid.reshape((B,) + (1,) * (parts[0][k].ndim - 1)) == 0, parts[0][k], parts[1][k]) for k in parts[0]}
    d["model_id"] = mid
    d["q"] = d["q"].copy()
    d["q"][6, 2] = np.nan                       # the base height
    d["q"][1001, 4] = np.inf                    # a quaternion component
    d["ee_target"] = d["ee_target"].copy()
    d["ee_target"][2007, 4, 1] = np.nan         # the gripper target
    ref = oracle.tick(models, cfgs, d, DT, B, nthreads=8)
    bad = np.zeros(B, bool)
    bad[[6, 1001, 2007]] = True
    assert (ref["status"][bad] != 0).all() and (ref["status"][~bad] == 0).mean() > 0.97
    bt = WbcBatch(models, B)
    for i, c in enumerate(cfgs):
        bt.configure(c, i)
    got = bt.tick(d, DT, want_q_next=True)
    assert bt.stat("last_path") == 3 and bt.stat("last_orth") == 1     # the default policy: packed from 4608 instances on
    ok = ref["status"] == 0
    mism = np.nonzero((got["status"] != ref["status"]) & ~bad)[0]
    assert mism.size == 0, (mism[:10], got["status"][mism[:10]], ref["status"][mism[:10]])     # infeasible trunk boxes included
    assert (got["status"][bad] != 0).all()                             # (which code a NaN ends in is the path's own business)
    assert (got["qdot"][~ok] == 0).all() and np.isfinite(got["qdot"]).all()
    err = np.abs(got["qdot"] - ref["qdot"])[ok].max()
    nd = bt.stat("deferred_last")
    print("packed orth kernel with inequality rows, %s, mixed batch: qdot max-abs err %.3e, working-set changes %.2f (oracle %.2f), %d instances redone in the tail, infeasible %d" % (
        variant, err, got["iters"][ok].mean(), ref["iters"][ok].mean(), nd, int((ref["status"] == 2).sum())))
    assert err < QDOT_TOL and np.abs(got["q_next"] - ref["q_next"])[ok].max() < 1e-7
    assert nd <= 0.02 * B
    assert (got["iters"] == ref["iters"])[ok & (mid == 0)].mean() > 0.98
    assert (got["iters"] + 1 == ref["iters"])[ok & (mid == 1)].mean() > 0.98   # (the oracle also counts px100's padded 26th DoF, a locked bound)
    if variant == "everything":   # a short closed-loop roll-out stays on the packed kernels too (cold ticks: the INEQ variant carries no working sets)
        K = 3
        clean = {k: (v[~bad] if isinstance(v, np.ndarray) and v.shape[0] == B else v) for k, v in d.items()}
        Bc = int((~bad).sum())
        rro = oracle.rollout(models, cfgs, clean, DT, Bc, K, nthreads=8)
        bt2 = WbcBatch(models, Bc)
        for i, c in enumerate(cfgs):
            bt2.configure(c, i)
        bt2.set_option("warm_start", 0)
        bt2.set_option("packed_orth", 2)
        gro = bt2.rollout(clean, DT, K)
        assert bt2.stat("last_path") == 3 and bt2.stat("last_update_packed") == 1
        okr = rro["status"] == 0
        assert (gro["status"] == rro["status"]).all() and np.abs(gro["q"] - rro["q"])[okr].max() < 1e-6
        bt2.close()
    bt.set_option("packed_orth", 0)
    one = bt.tick(d, DT)
    assert bt.stat("last_path") == 0 and np.abs(one["qdot"] - got["qdot"])[ok].max() < 1e-6 and (one["status"] == got["status"])[~bad].all()
    bt.close()


@pytest.mark.parametrize("variant", ["warm_up", "prev_no_trunk", "three_tasks"])
def test_packed_box_kernel_variants_and_non_finite_inputs(wx200, px100, variant):
    """wbc_tick_boxp_kernel (task problems without constraint rows, four per wavefront, base + one thigh eliminated by a Schur complement):
    the warm-up problem of setInitialState (Robot_Wrapper4.py:196-351: six Cartesian tasks + Tikhonov) with moving orientation references,
    the PREV posture without the trunk task, and a stack in which two legs and the arm carry no task at all (their DoF see the posture row
    only) — each on a mixed wx200 / px100 batch at a size the default policy sends to it, ragged tail, q_next, working-set changes equal to
    the oracle's (the reduced problem's dual iterates ARE the full problem's), and non-finite inputs contained to their own instance."""
    B = 4610
    models = [wx200, px100]
    kw = dict(warm_up=dict(Trunk=True, FR=True, FL=True, RR=True, RL=True, Grip=True, Joint=True),
              prev_no_trunk=dict(FR=True, FL=True, RR=True, RL=True, Grip=True, Joint="PREV"),
              three_tasks=dict(Trunk=True, FR=True, RR=True, Joint=True))[variant]
    cfgs = [wbc_model.make_config(m, **kw) for m in models]
    mid = (np.arange(B) % 2).astype(np.int32)
    parts = [common.tick_inputs(m, c, B, seed=91 + i, with_rot=(variant != "three_tasks")) for i, (m, c) in enumerate(zip(models, cfgs))]
    d = {k: np.where(mid.reshape((B,) + (1,) * (parts[0][k].ndim - 1)) == 0, parts[0][k], parts[1][k]) for k in parts[0]}
    d["model_id"] = mid
    d["q"] = d["q"].copy()
    d["q"][6, 2] = np.nan                       # the base height
    d["q"][1001, 4] = np.inf                    # a quaternion component
    d["ee_target"] = d["ee_target"].copy()
    d["ee_target"][2007, 0, 1] = np.nan         # the FR foot's target
    ref = oracle.tick(models, cfgs, d, DT, B, nthreads=8)
    bad = np.zeros(B, bool)
    bad[[6, 1001, 2007]] = True
    assert (ref["status"][bad] != 0).all() and (ref["status"][~bad] == 0).mean() > 0.99
    bt = WbcBatch(models, B)
    for i, c in enumerate(cfgs):
        bt.configure(c, i)
    assert bt.constraint_rows == 0
    got = bt.tick(d, DT, want_q_next=True)
    assert bt.stat("last_path") == 4                                   # the default policy: the packed box kernel at every batch size (WBC_BOXP_MIN_BATCH = 1)
    ok = ref["status"] == 0
    # the tail takes exactly the instances whose optimum holds an ELIMINATED DoF (base; wx200: + the first thigh) at its velocity bound: none in the
    # warm-up problem (the trunk task pins the base), a handful without the trunk task
    asm = oracle.assemble(models, cfgs, d, DT, B)
    elim = np.zeros((B, 26), bool)
    elim[:, :6] = True
    elim[mid == 0, 7] = True
    held = ((np.abs(ref["qdot"] - asm["lb"]) < 1e-9 * np.maximum(1, np.abs(asm["lb"]))) | (np.abs(ref["qdot"] - asm["ub"]) < 1e-9 * np.maximum(1, np.abs(asm["ub"])))) & elim
    expect = int((held.any(axis=1) & ok).sum())
    nd = bt.stat("deferred_last")
    print("   instances redone in the tail: %d (oracle: %d with an eliminated DoF at its bound)" % (nd, expect))
    assert abs(nd - expect) <= 2 and nd <= 0.02 * B and (variant != "warm_up" or nd == 0)
    assert ((got["status"] != 0) == (ref["status"] != 0)).all()
    assert (got["qdot"][~ok] == 0).all() and np.isfinite(got["qdot"]).all()
    err = np.abs(got["qdot"] - ref["qdot"])[ok].max()
    print("packed box kernel, %s, mixed batch: qdot max-abs err %.3e, working-set changes %.2f (oracle %.2f)" % (
        variant, err, got["iters"][ok].mean(), ref["iters"][ok].mean()))
    assert err < (1e-8 if nd == 0 else QDOT_TOL) and np.abs(got["q_next"] - ref["q_next"])[ok].max() < 1e-8    # (the Schur complement conditions the problem: far inside QDOT_TOL)
    assert (got["iters"] == ref["iters"])[ok & (mid == 0)].mean() > 0.99
    assert (got["iters"] + 1 == ref["iters"])[ok & (mid == 1)].mean() > 0.99   # (the oracle also counts px100's padded 26th DoF, a locked bound)
    bt.set_option("packed_box", 0)
    one = bt.tick(d, DT)
    assert bt.stat("last_path") == 0 and np.abs(one["qdot"] - got["qdot"])[ok].max() < 1e-8 and (one["status"] == got["status"]).all()
    bt.close()


@pytest.mark.parametrize("case", ["base_bound", "many_bounds"])
def test_packed_box_kernel_redoes_what_its_reduction_does_not_cover(wx200, case):
    """The packed box kernel eliminates DoF whose velocity bounds it only CHECKS afterwards, and holds at most 12 active bounds: an instance
    whose optimum needs an eliminated DoF at its bound (here: the base box shrunk to 0.05 — vel_lim[0:6] = 5 in the reference never binds)
    or more active bounds than that (every leg / arm limit shrunk to 0.02) is redone by its own wave on the general path — same answer as
    the oracle, statistic "deferred_last" counts them, the other instances of the wavefront are untouched."""
    B = 515
    cfg = common.config("full", wx200)
    if case == "base_bound":
        for i in range(6):
            cfg.damper_vmax[i] = 0.05
    else:
        for i in range(6, 23):
            cfg.damper_vmax[i] = 0.02
    d = common.tick_inputs(wx200, cfg, B, seed=93, with_rot=True)
    ref = oracle.tick([wx200], [cfg], d, DT, B, nthreads=8)
    ok = ref["status"] == 0
    assert ok.mean() > 0.95
    lim = 0.05 if case == "base_bound" else 0.02
    at_bound = (np.abs(np.abs(ref["qdot"][:, :6] if case == "base_bound" else ref["qdot"][:, 6:23]) - lim) < 1e-12).sum(axis=1)
    expect = (at_bound > 0) if case == "base_bound" else (at_bound > 12)
    assert 0.1 < expect[ok].mean()
    bt = WbcBatch(wx200, B)
    bt.configure(cfg)
    bt.set_option("packed_box", 2)
    got = bt.tick(d, DT, want_q_next=True)
    assert bt.stat("last_path") == 4
    nd = bt.stat("deferred_last")
    print("%s: %d of %d instances took the tail (oracle: %d with %s)" % (case, nd, B, expect[ok].sum(), "a base DoF at its bound" if case == "base_bound" else "more than 12 active bounds"))
    assert nd >= expect[ok].sum() and (case == "many_bounds" or nd == expect[ok].sum())
    assert (got["status"] == ref["status"]).all()
    assert np.abs(got["qdot"] - ref["qdot"])[ok].max() < QDOT_TOL and np.abs(got["q_next"] - ref["q_next"])[ok].max() < 1e-7
    assert (got["iters"] == ref["iters"])[ok].mean() > 0.98
    bt.close()


def test_rollouts_are_deterministic_and_survive_unsolvable_ticks(wx200):
    """Run-to-run bit equality of a long roll-out in which some instances run into unsolvable QPs on the way (they hold
    still from then on instead of integrating a partial iterate into garbage — found with tools/determinism.py)."""
    B, K = 8192, 50
    cfg = common.config("c3", wx200)
    d = common.tick_inputs(wx200, cfg, B, seed=3)
    step = np.zeros((B, 5, 3))
    step[:, 4, 0] = 1e-4
    bt = WbcBatch(wx200, B)
    bt.configure(cfg)
    r1 = bt.rollout(d, DT, K, ee_target_step=step)
    bt.fk(d["q"][:777])                                   # different LDS leftovers in between
    r2 = bt.rollout(d, DT, K, ee_target_step=step)
    for k in r1:
        assert np.array_equal(r1[k], r2[k]), k
    assert np.isfinite(r1["q"]).all() and np.abs(r1["q"][:, 7:]).max() < 10.0     # nobody left the planet
    assert (r1["status"] != 0).sum() > 0                                          # the sample does contain unsolvable ticks
    bt.close()


@pytest.mark.parametrize("cfg_name,with_rot,model_name", [("c3", False, "wx200"), ("c3", False, "px100"), ("c2", False, "wx200"),
                                                          ("everything", True, "wx200"), ("hybrid_grip_com", False, "wx200"),
                                                          ("full", True, "wx200"), ("full", True, "px100")])
def test_gpu_solution_against_the_exact_optimum(wx200, px100, cfg_name, with_rot, model_name):
    """The device's q̇ against the EXACT (rational-arithmetic) optimum of the double-precision QP data it assembled itself: no oracle
    in between (tests/common.py exact_optimum: Gaussian elimination in fractions on the active set the answer shows, optimality
    conditions asserted). Every kernel path the configuration can take — packed / one-instance compact / general with the explicit or
    the orthonormal contact presolve / general at full size — must be as close to the truth as the oracle is."""
    model = wx200 if model_name == "wx200" else px100
    cfg = common.config(cfg_name, model)
    B = 24
    d = common.tick_inputs(model, cfg, B, seed=71, with_rot=with_rot)
    bt = WbcBatch(model, B)
    bt.configure(cfg)
    a = bt.assemble(d, DT)
    nv = model.nv
    exact, worst, status0 = {}, {}, None
    # (presolve, presolve_orth, sim3_kernel, packed_kernel)
    for key in ((1, 1, 1, 1), (1, 1, 1, 0), (1, 1, 0, 0), (1, 0, 0, 0), (0, 0, 0, 0)):
        for name, v in zip(("presolve", "presolve_orth", "sim3_kernel", "packed_kernel"), key):
            bt.set_option(name, v)
        bt.set_option("refine", 0 if key == (1, 1, 1, 0) else 1)     # (the one-instance compact kernel: chosen with the refinement off only)
        bt.set_option("packed_orth", 2)      # (small batch: 1 would keep config 2 on the one-instance kernel)
        bt.set_option("packed_box", 2 if key[3] else 0)
        got = bt.tick(d, DT)
        path = (bt.stat("last_path"), bt.stat("last_orth"))
        # an instance is solved on every kernel path or on none: a path that fails where another succeeds must not go unnoticed
        if status0 is None:
            status0 = got["status"].copy()
        assert (got["status"] == status0).all(), "status differs between kernel paths %s: %s vs %s" % (key, got["status"], status0)
        for b in range(B):
            if got["status"][b] != 0:
                continue
            if b not in exact:
                # the exact optimum of the least-squares problem the device assembled (H = A'A, g = -A'b in rational arithmetic from its A, b)
                exact[b] = common.exact_ls_optimum(a["A"][b][:, :nv], a["b"][b], a["C"][b][:, :nv], a["lb"][b][:nv], a["ub"][b][:nv],
                                                   a["Clb"][b], a["Cub"][b], got["qdot"][b][:nv])
            err = np.abs(got["qdot"][b][:nv] - exact[b]).max()
            worst[(key, path)] = max(worst.get((key, path), 0.0), err)
    print(cfg_name, model_name, {k: "%.2e" % v for k, v in worst.items()}, "%d instances" % len(exact))
    assert len(exact) >= B - 4
    # every path within 1e-7 of the exact optimum (refined paths: 1e-8 class; the un-refined ones — orthonormal presolve / packed orth kernel on the
    # well-conditioned CoM-task stacks — 1e-8 too); only the one-instance compact kernel, which has no room for the refinement, keeps round 3's 2e-6
    for (key, path), w in worst.items():
        assert w < (2e-6 if path == (1, 0) else 1e-7), (key, path, w)
    paths = {k[1] for k in worst}
    if cfg_name == "c3":
        assert {(2, 0), (1, 0), (0, 0)} <= paths          # packed, one-instance compact, general
    if cfg_name in ("c2", "everything"):
        assert {(0, 1), (0, 0)} <= paths                  # orthonormal presolve, full size
    if cfg_name in ("c2", "everything"):
        assert (3, 1) in paths                            # the packed orth kernel (configs[1]) / its INEQ variant ("everything")
    if cfg_name == "full":
        assert {(4, 0), (0, 0)} <= paths                  # the packed box kernel, full size
        assert worst[((1, 1, 1, 1), (4, 0))] < 1e-9      # the Schur complement conditions the problem: the packed path is the MOST accurate one
    bt.close()


@pytest.mark.parametrize("B", [1, 63, 64, 65, 1000])
def test_ragged_batch_sizes(wx200, B):
    """Batch sizes around the 64-status granularity of the deferred pass and a handle larger than the batch: every
    instance is solved, nothing beyond B is touched."""
    cfg = common.config("c3", wx200)
    d = common.tick_inputs(wx200, cfg, B, seed=90 + B)
    ref = oracle.tick([wx200], [cfg], d, DT, B, nthreads=8)
    bt = WbcBatch(wx200, 1024)                       # max_batch > B
    bt.configure(cfg)
    guard = 7
    out = dict(qdot=np.full((B + guard, 26), -77.0), status=np.full(B + guard, -5, dtype=np.int32),
               iters=np.full(B + guard, -5, dtype=np.int32), q_next=np.full((B + guard, 27), -77.0))
    views = {k: v[:B] for k, v in out.items()}
    got = bt.tick(d, DT, out=views)
    assert (got["status"] == ref["status"]).all()
    ok = ref["status"] == 0
    assert np.abs(got["qdot"] - ref["qdot"])[ok].max() < QDOT_TOL
    assert (out["qdot"][B:] == -77.0).all() and (out["status"][B:] == -5).all() and (out["q_next"][B:] == -77.0).all()
    bt.close()


def test_tick_can_be_captured_into_a_graph(wx200):
    """include/wbc.h: with device pointers a call only enqueues kernels, so after one warm-up call it can be captured on
    the caller's stream (here through torch's HIP-graph wrapper) and replayed."""
    import torch
    B = 512
    cfg = common.config("c3", wx200)
    d = common.tick_inputs(wx200, cfg, B, seed=14)
    bt = WbcBatch(wx200, B)
    bt.configure(cfg)
    dev = torch.device("cuda", 0)
    dd = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in d.items()}
    out = dict(qdot=torch.zeros((B, 26), dtype=torch.float64, device=dev), status=torch.zeros(B, dtype=torch.int32, device=dev),
               iters=torch.zeros(B, dtype=torch.int32, device=dev), q_next=torch.zeros((B, 27), dtype=torch.float64, device=dev))
    step = bt.make_tick_call(dd, out, DT)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        step()                                   # warm-up: workspaces are allocated here, not during capture
    side.synchronize()
    ref = {k: v.clone() for k, v in out.items()}
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        step()
    for v in out.values():
        v.zero_()
    g.replay()
    torch.cuda.synchronize()
    for k in out:
        assert torch.equal(out[k], ref[k]), k
    dd["q"][:, 19] += 0.01                       # new inputs in the same buffers: the replay follows them
    g.replay()
    torch.cuda.synchronize()
    assert not torch.equal(out["qdot"], ref["qdot"])
    bt.close()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("cfg_name", ["c3", "c2"])
def test_non_finite_inputs_are_contained(wx200, cfg_name):
    """NaN / Inf / absurd magnitudes in some instances' inputs: every wave still terminates (all solver loops are capped),
    the poisoned instances come back flagged with q̇ = 0, and their neighbours are solved as if nothing had happened — on the packed /
    compact sim3 kernels (c3) and through the orthonormal contact presolve of the general kernel (c2)."""
    B = 512
    cfg = common.config(cfg_name, wx200)
    d = common.tick_inputs(wx200, cfg, B, seed=33)
    clean = {k: v.copy() for k, v in d.items()}
    d["q"][5, 10] = np.nan
    d["q"][17, 2] = np.inf
    d["q"][40, 3:7] = 0.0                      # zero quaternion
    d["ee_target"][63, 4, 1] = np.nan
    d["ee_target"][64, 4, 0] = 1e200
    d["trunk_box_center"][100, 0] = -np.inf if cfg_name == "c3" else d["trunk_box_center"][100, 0]
    if cfg_name == "c2":
        d["com_target"][100, 1] = np.inf
    d["q"][200, 20] = 1e30                     # an angle no range reduction survives
    poisoned = [5, 17, 40, 63, 64, 100, 200]
    bt = WbcBatch(wx200, B)
    bt.configure(cfg)
    for sim3 in (1, 0):
        bt.set_option("sim3_kernel", sim3)
        bt.set_option("presolve", sim3)
        ref = bt.tick(clean, DT, want_q_next=True)
        got = bt.tick(d, DT, want_q_next=True)
        keep = np.ones(B, dtype=bool)
        keep[poisoned] = False
        assert np.array_equal(got["qdot"][keep], ref["qdot"][keep]) and np.array_equal(got["status"][keep], ref["status"][keep])
        for i in poisoned:
            assert got["status"][i] != 0 or np.isfinite(got["qdot"][i]).all(), i
            if got["status"][i] != 0:
                assert (got["qdot"][i] == 0).all(), i
        assert (got["status"][[5, 17, 63, 100]] != 0).all()      # NaN / Inf cannot produce an "optimal" answer
        bt.set_option("sim3_kernel", 1)
        bt.set_option("presolve", 1)
    bt.close()


def test_stray_model_indices_are_clamped(wx200, px100):
    """A model index outside the handle's models in the caller's buffer is clamped, not followed into an out-of-bounds
    table read (device buffers cannot be validated on the host)."""
    B = 64
    models = [wx200, px100]
    cfgs = [common.config("c3", m) for m in models]
    d = common.tick_inputs(wx200, cfgs[0], B, seed=8)
    mid = np.zeros(B, dtype=np.int32)
    bad = mid.copy()
    bad[3], bad[9] = -7, 1_000_000
    want = mid.copy()
    want[9] = 1
    bt = WbcBatch(models, B)
    for i, c in enumerate(cfgs):
        bt.configure(c, i)
    a = bt.tick(dict(d, model_id=bad), DT, want_q_next=True)
    b_ = bt.tick(dict(d, model_id=want), DT, want_q_next=True)
    for k in a:
        assert np.array_equal(a[k], b_[k]), k
    assert np.array_equal(bt.fk(d["q"], bad)["oMf"], bt.fk(d["q"], want)["oMf"])
    assert np.array_equal(bt.integrate(d["q"], a["qdot"], DT, bad), bt.integrate(d["q"], a["qdot"], DT, want))
    bt.close()


def test_config4_as_eight_shards_on_one_gpu(wx200):
    """BASELINE configs[3]: B = 524288 = 8 x 65536 shards, one per GPU, no communication. On one GPU: the 8 contiguous
    wbc_shard.shard_range shards go one after the other through a 65536-instance handle (what each rank of the 8-GPU run does
    with its own shard); their union must be bit-identical to ONE monolithic 524288-instance call, and match the oracle on a
    2048-instance subsample."""
    import wbc_shard
    N, W = 524288, 8
    cfg = common.config("c3", wx200)
    d = common.tick_inputs(wx200, cfg, N, seed=404)
    union = dict(qdot=np.empty((N, 26)), status=np.empty(N, dtype=np.int32), iters=np.empty(N, dtype=np.int32), q_next=np.empty((N, 27)))
    bt = WbcBatch(wx200, N // W)
    bt.configure(cfg)
    for r in range(W):
        lo, hi = wbc_shard.shard_range(N, r, W)
        assert hi - lo == 65536
        got = bt.tick({k: v[lo:hi] for k, v in d.items()}, DT, want_q_next=True)
        for k in union:
            union[k][lo:hi] = got[k]
    bt.close()
    big = WbcBatch(wx200, N)
    big.configure(cfg)
    mono = big.tick(d, DT, want_q_next=True)
    big.close()
    for k in union:
        assert np.array_equal(union[k], mono[k]), k
    assert (union["status"] == 0).mean() > 0.98
    rng = np.random.default_rng(1)
    idx = rng.choice(N, 2048, replace=False)
    ref = oracle.tick([wx200], [cfg], {k: v[idx] for k, v in d.items()}, DT, len(idx), nthreads=8)
    assert (ref["status"] == union["status"][idx]).all()
    good = ref["status"] == 0
    assert np.abs(ref["qdot"] - union["qdot"][idx])[good].max() < QDOT_TOL
    assert np.abs(ref["q_next"] - union["q_next"][idx])[good].max() < 1e-7


def test_mixed_morphology_full_size_properties(wx200, px100):
    """BASELINE configs[4] (mixed wx200 / px100 lanes) at the full single-GPU batch B = 65536: contact rows satisfied, box
    rows and bounds respected on every instance, padded DoF untouched, oracle parity on a 1024-instance subsample."""
    B = 65536
    models = [wx200, px100]
    cfgs = [common.config("c3", m) for m in models]
    mid = (np.arange(B) % 2).astype(np.int32)
    parts = [common.tick_inputs(m, c, B, seed=57 + i) for i, (m, c) in enumerate(zip(models, cfgs))]
    d = {k: np.where(mid.reshape((B,) + (1,) * (parts[0][k].ndim - 1)) == 0, parts[0][k], parts[1][k]) for k in parts[0]}
    d["model_id"] = mid
    bt = WbcBatch(models, B)
    for i, c in enumerate(cfgs):
        bt.configure(c, i)
    got = bt.tick(d, DT)
    assert bt.stat("last_path") == 2                                   # the packed compact kernel, both morphologies in one wavefront
    a = bt.assemble(d, DT, want=("C", "Clb", "Cub", "lb", "ub"))
    ok = got["status"] == 0
    assert ok.mean() > 0.98
    x = got["qdot"]
    Cx = np.einsum("bpn,bn->bp", a["C"], x)
    scale = 1 + np.abs(x).max(axis=1, keepdims=True)
    assert (np.abs(Cx[:, 4:])[ok] / scale[ok]).max() < 1e-8
    assert ((a["Clb"] - Cx)[ok] / scale[ok]).max() < 1e-8 and ((Cx - a["Cub"])[ok] / scale[ok]).max() < 1e-8
    assert ((a["lb"] - x)[ok]).max() < 1e-8 and ((x - a["ub"])[ok]).max() < 1e-8
    assert (x[mid == 1, 25] == 0).all()
    rng = np.random.default_rng(5)
    idx = np.sort(rng.choice(B, 1024, replace=False))
    ref = oracle.tick(models, cfgs, {k: v[idx] for k, v in d.items()}, DT, len(idx), nthreads=8)
    assert (ref["status"] == got["status"][idx]).all()
    good = ref["status"] == 0
    assert np.abs(ref["qdot"] - x[idx])[good].max() < QDOT_TOL
    bt.close()


def test_batched_warm_up_matches_the_oracle(wx200, px100):
    """setInitialState (Robot_Wrapper4.py:196-351) for a mixed batch in ONE wbc_rollout call
    (mode WBC_ROLLOUT_WARMUP, 1000 moving + 1000 held ticks = 2000 bounds-only QPs per robot) against oracle.warmup and
    against the committed fixture. State tolerance as in test_rollout_parity (targets accumulate by steps on the device,
    the oracle evaluates the reference's piecewise-linear interpolation: ~1e-13 apart per tick)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "warmup_mixed.npz"))
    models = [wx200, px100]
    q0, mid, fr = z["q0"], z["model_id"], float(z["foot_radius"])
    B = q0.shape[0]
    bt = WbcBatch(models, B)
    got = bt.warm_up(q0, mid, DT, int(z["ticks_per_segment"]), foot_radius=fr)
    assert (got["status"] == z["out_status"]).all() and (got["status"] == 0).all()
    assert np.abs(got["q"] - z["out_q"]).max() < 1e-6
    assert np.abs(got["goal"] - z["out_goal"]).max() < 1e-12 and np.abs(got["start"] - z["out_start"]).max() < 1e-12
    # a short warm-up straight against the oracle (fresh, not from the fixture)
    ref = oracle.warmup(models, q0, DT, 50, foot_radius=fr, model_id=mid, nthreads=8)
    short = bt.warm_up(q0, mid, DT, 50, foot_radius=fr)
    assert (short["status"] == ref["status"]).all()
    assert np.abs(short["q"] - ref["q"]).max() < 1e-6
    # working-set changes: equal for wx200; the oracle also counts px100's padded 26th DoF (a locked bound) once per tick
    assert np.abs(short["iters"] - ref["iters"])[mid == 0].max() <= 4
    assert np.abs(short["iters"] + 100 - ref["iters"])[mid == 1].max() <= 4
    # the same warm-up with every tick on the packed box kernel (the default policy: at every batch size) and the packed state update
    bt.set_option("packed_box", 2)
    packed = bt.warm_up(q0, mid, DT, 50, foot_radius=fr)
    assert bt.stat("last_path") == 4 and bt.stat("last_update_packed") == 1
    assert (packed["status"] == ref["status"]).all() and np.abs(packed["q"] - ref["q"]).max() < 1e-6
    assert np.abs(packed["iters"] - ref["iters"])[mid == 0].max() <= 4
    bt.close()


def test_rollout_warmup_mode_is_tick_plus_integrate(wx200):
    """WBC_ROLLOUT_WARMUP = updateState(new_config, feedback=False, running=False): the integrated configuration is the next
    state, bit for bit what chaining wbc_tick (+ q_next) from the host gives; hold_ticks keeps the targets in place."""
    B, K, Hd = 32, 3, 2
    cfg = common.config("full", wx200)
    d = common.tick_inputs(wx200, cfg, B, seed=47, with_rot=True)
    step = np.zeros((B, 5, 3))
    step[:, :, 2] = -1e-4
    bt = WbcBatch(wx200, B)
    bt.configure(cfg)
    bt.set_option("warm_start", 0)
    got = bt.rollout(d, DT, K, ee_target_step=step, mode=capi.ROLLOUT_WARMUP, hold_ticks=Hd)
    assert got["grip_trace"].shape == (K + Hd, B, 3)
    s = {k: v.copy() for k, v in d.items()}
    for k in range(K + Hd):
        o = bt.tick(s, DT, want_q_next=True)
        s["q"] = o["q_next"]
        s["prev_ee_target"] = s["ee_target"].copy()
        s["ee_prev_rot"] = s["ee_ref_rot"].copy()
        s["prev_trunk_target"] = s["trunk_target"].copy()
        from scipy.spatial.transform import Rotation as R
        s["trunk_prev_rot"] = R.from_euler("xyz", s["trunk_ref_euler"]).as_matrix().reshape(B, 9)
        if k < K:
            s["ee_target"] = s["ee_target"] + step
    assert np.abs(got["q"] - s["q"]).max() < 1e-12 and (got["ee_target"] == s["ee_target"]).all()
    assert np.abs(got["ee_target"] - (d["ee_target"] + K * step)).max() < 1e-15
    bt.close()


def test_qp_entry_points_take_and_return_working_sets():
    """wbc_qp_solve / wbc_qp_solve_ls with working_set_in / _out (QP.solveQPHotstart, QP_Wrapper.py:55-73): seeded with its own final set, a
    perturbed problem's, garbage or nothing, the answer is the oracle's; the own set costs one step per active constraint."""
    rng = np.random.default_rng(21)
    B, n, p, m = 256, 14, 9, 20
    A = rng.normal(size=(B, m, n))
    b = rng.normal(size=(B, m)) * 3
    H = np.einsum("bmi,bmj->bij", A, A) + 1e-3 * np.eye(n)
    g = -np.einsum("bmi,bm->bi", A, b)
    C = rng.normal(size=(B, p, n))
    lb, ub = -rng.uniform(0.05, 0.6, (B, n)), rng.uniform(0.05, 0.6, (B, n))
    cl, cu = -rng.uniform(0.05, 0.6, (B, p)), rng.uniform(0.05, 0.6, (B, p))
    cl[:, 0] = cu[:, 0] = 0.02                                      # an equality row
    xr, sr, ir = oracle.qp_solve(H, g, C, lb, ub, cl, cu)
    ok = sr == 0
    assert ok.mean() > 0.9
    bt = WbcBatch([], B)
    x0, s0, i0, ws0 = bt.qp_solve(H, g, C, lb, ub, cl, cu, want_working_set=True)
    assert (s0 == sr).all() and np.abs(x0 - xr)[ok].max() < 1e-8 and (i0 == ir)[ok].all()
    assert (ws0[~ok] == 0).all() and (ws0[ok] != 0).any()
    nact = np.array([bin(int(w) & (2 ** 64 - 1)).count("1") for w in ws0.ravel()]).reshape(B, 2).sum(axis=1)
    x1, s1, i1, ws1 = bt.qp_solve(H, g, C, lb, ub, cl, cu, working_set=ws0, want_working_set=True)
    assert (s1 == sr).all() and np.abs(x1 - xr)[ok].max() < 1e-8 and (ws1 == ws0)[ok].all()
    # one step per seed on top of the equality row where every seed is taken; the filter turns far seeds away (they come back through dual
    # iterations) on these random problems with ~10 active constraints — never more changes than cold on average
    assert i1[ok].mean() <= i0[ok].mean(), (i1[ok].mean(), i0[ok].mean())
    print("own set: %.0f %% of the QPs need exactly one step per seed" % (100 * (i1[ok] <= 1 + nact[ok]).mean()))
    junk = rng.integers(-2 ** 62, 2 ** 62, (B, 2), dtype=np.int64)
    x2, s2, _ = bt.qp_solve(H, g, C, lb, ub, cl, cu, working_set=junk)
    assert (s2 == sr).all() and np.abs(x2 - xr)[ok].max() < 1e-7
    # least-squares form, seeded with the set of a perturbed right-hand side (the "previous tick"), in place
    _, sp, _, wsp = bt.qp_solve_ls(A, b + rng.normal(size=(B, m)) * 0.1, C, lb, ub, cl, cu, want_working_set=True)
    x3, s3, i3, ws3 = bt.qp_solve_ls(A, b, C, lb, ub, cl, cu, working_set=wsp, want_working_set=True)
    Hl, gl = np.einsum("bmi,bmj->bij", A, A), g
    xl, sl, il = oracle.qp_solve(Hl, gl, C, lb, ub, cl, cu)
    okl = sl == 0
    assert (s3 == sl).all() and np.abs(x3 - xl)[okl].max() < 1e-6
    print("working-set changes per QP: cold %.2f, own set %.2f, previous problem's set %.2f" % (i0[ok].mean(), i1[ok].mean(), i3[okl].mean()))
    bt.close()


@pytest.mark.parametrize("m,n,p,lanes", [(32, 26, 16, 2), (20, 14, 9, 4), (36, 26, 20, 2)])      # (split rows of C; four per wavefront; whole rows: p > 16)
def test_qp_hot_start_on_the_packed_kernel(m, n, p, lanes):
    """QP.solveQPHotstart (QP_Wrapper.py:55-73) on the packed kernel's WARM variant (csrc/wbc_k_qpp.hip): seeded with its own final set the problem
    comes back with the same answer, the same set and no more working-set changes than cold; with the set of a perturbed problem ("previous tick")
    or garbage the answer is the oracle's; the one-per-wavefront kernel (option packed_kernel 0) returns the same sets bit for bit."""
    rng = np.random.default_rng(70 + n)
    B = 257
    A = rng.normal(size=(B, m, n))
    b = rng.normal(size=(B, m)) * 3
    C = rng.normal(size=(B, p, n))
    lb, ub = -rng.uniform(0.05, 0.6, (B, n)), rng.uniform(0.05, 0.6, (B, n))
    lb[:, 2] = ub[:, 2] = 0.01                                      # a fixed variable: presolved, never in a carried set
    cl, cu = -rng.uniform(0.05, 0.6, (B, p)), rng.uniform(0.05, 0.6, (B, p))
    cl[:, 1] = cu[:, 1] = 0.02                                      # an equality row: always in, never carried
    xr, sr, ir = oracle.qp_solve_ls(A, b, C, lb, ub, cl, cu)
    ok = sr == 0
    assert ok.mean() > 0.9
    bt = WbcBatch([], B)
    x0, s0, i0, ws0 = bt.qp_solve_ls(A, b, C, lb, ub, cl, cu, want_working_set=True)
    assert bt.stat("last_qp_path") == lanes
    assert (s0 == sr).all() and np.abs(x0 - xr)[ok].max() < 1e-8 and (i0 == ir)[ok].all()
    assert (ws0[~ok] == 0).all() and (ws0[ok] != 0).any()
    assert ((ws0[:, 0] >> 2) & 1 == 0).all() and ((ws0[:, 0] >> 34) & 1 == 0).all() and ((ws0[:, 1] >> 1) & 1 == 0).all() and ((ws0[:, 1] >> 33) & 1 == 0).all()
    x1, s1, i1, ws1 = bt.qp_solve_ls(A, b, C, lb, ub, cl, cu, working_set=ws0, want_working_set=True)
    assert bt.stat("last_qp_path") == lanes
    assert (s1 == sr).all() and np.abs(x1 - xr)[ok].max() < 1e-8 and (ws1 == ws0)[ok].all()
    assert i1[ok].mean() <= i0[ok].mean(), (i1[ok].mean(), i0[ok].mean())
    _, _, _, wsp = bt.qp_solve_ls(A, b + rng.normal(size=(B, m)) * 0.1, C, lb, ub, cl, cu, want_working_set=True)
    x2, s2, i2, ws2 = bt.qp_solve_ls(A, b, C, lb, ub, cl, cu, working_set=wsp, want_working_set=True)
    assert (s2 == sr).all() and np.abs(x2 - xr)[ok].max() < 1e-7 and (ws2 == ws0)[ok].all()
    junk = rng.integers(-2 ** 62, 2 ** 62, (B, 2), dtype=np.int64)
    x3, s3, _ = bt.qp_solve_ls(A, b, C, lb, ub, cl, cu, working_set=junk)
    assert (s3 == sr).all() and np.abs(x3 - xr)[ok].max() < 1e-7
    print("working-set changes per QP: cold %.2f, own set %.2f, previous problem's set %.2f" % (i0[ok].mean(), i1[ok].mean(), i2[ok].mean()))
    bt.set_option("packed_kernel", 0)
    x4, s4, i4, ws4 = bt.qp_solve_ls(A, b, C, lb, ub, cl, cu, working_set=wsp, want_working_set=True)
    assert bt.stat("last_qp_path") == 1
    assert (s4 == sr).all() and (ws4 == ws2)[ok].all() and np.abs(x4 - x2)[ok].max() < 1e-8
    bt.close()


@pytest.mark.parametrize("cfg_name", ["c3", "everything", "c2", "full"])
def test_warm_started_tick_reaches_the_cold_optimum(wx200, px100, cfg_name):
    """SURVEY.md §8 f2 (QP_Wrapper.py:55-73, Robot_Wrapper4.py:1389-1394): a tick seeded with a working set — the previous
    tick's, its own, another kernel's, or garbage — returns the cold tick's q̇ and status (H > 0: one minimiser), and a good
    seed costs fewer working-set changes than the cold dual iterations."""
    B = 2048
    models = [wx200, px100]
    cfgs = [common.config(cfg_name, m) for m in models]
    mid = (np.arange(B) % 2).astype(np.int32)
    parts = [common.tick_inputs(m, c, B, seed=91 + i, with_rot=(cfg_name in ("everything", "full"))) for i, (m, c) in enumerate(zip(models, cfgs))]
    d = {k: np.where(mid.reshape((B,) + (1,) * (parts[0][k].ndim - 1)) == 0, parts[0][k], parts[1][k]) for k in parts[0]}
    d["model_id"] = mid
    ref = oracle.tick(models, cfgs, d, DT, B, nthreads=8)
    ok = ref["status"] == 0
    bt = WbcBatch(models, B)
    for i, c in enumerate(cfgs):
        bt.configure(c, i)
    if cfg_name == "everything":
        bt.set_option("packed_orth", 2)                                # (B = 2048: the default policy would keep it on the one-instance kernel) -> the INEQ variant, WARM
    # "previous tick": the same robots a moment earlier (targets 0.3 mm back)
    prev = dict(d)
    prev["ee_target"] = d["ee_target"] - 3e-4
    p0 = bt.tick(prev, DT, want_working_set=True)
    cold = bt.tick(d, DT, want_working_set=True)
    assert (cold["status"] == ref["status"]).all()
    assert (cold["working_set"][~ok] == 0).all()                          # an unsolved QP carries nothing
    nact = np.array([bin(int(w) & (2 ** 64 - 1)).count("1") for w in cold["working_set"].ravel()]).reshape(B, 2).sum(axis=1)
    rng = np.random.default_rng(3)
    junk = rng.integers(-2 ** 62, 2 ** 62, (B, 2), dtype=np.int64)
    opposite = np.stack([((w & 0xFFFFFFFF) << 32) | ((w >> 32) & 0xFFFFFFFF) for w in cold["working_set"].T], axis=1)
    runs = {"previous tick": p0["working_set"], "own": cold["working_set"], "garbage": junk, "opposite sides": opposite}
    its = {"cold": cold["iters"][ok].mean()}
    for name, ws in runs.items():
        got = bt.tick(dict(d, working_set=ws), DT, want_working_set=True)
        assert (got["status"] == ref["status"]).all(), name
        assert np.abs(got["qdot"] - ref["qdot"])[ok].max() < QDOT_TOL, name
        assert np.abs(got["qdot"] - cold["qdot"])[ok].max() < QDOT_TOL, name
        its[name] = got["iters"][ok].mean()
        if name == "own":
            assert (got["working_set"][ok] == cold["working_set"][ok]).mean() > 0.98
    # the working set means the same constraints on every kernel path: general-path sets seed the compact kernel and back
    if cfg_name == "c3":
        assert bt.stat("last_path") == 2                               # a working set was passed: the WARM packed kernel (four instances per wavefront)
        bt.set_option("packed_kernel", 0)                              # ... and the warm one-instance compact kernel names the same constraints
        bt.set_option("refine", 0)                                     # (chosen with the refinement off only)
        for name in ("own", "garbage"):
            one = bt.tick(dict(d, working_set=runs[name]), DT, want_working_set=True)
            assert bt.stat("last_path") == 1
            assert (one["status"] == ref["status"]).all() and np.abs(one["qdot"] - ref["qdot"])[ok].max() < QDOT_TOL, name
            assert (one["working_set"][ok] == cold["working_set"][ok]).all(axis=1).mean() > 0.98, name
        bt.set_option("refine", 1)
        bt.set_option("packed_kernel", 1)
        bt.set_option("sim3_kernel", 0)
        bt.set_option("presolve", 0)
        gen = bt.tick(dict(d, working_set=cold["working_set"]), DT, want_working_set=True)
        assert np.abs(gen["qdot"] - ref["qdot"])[ok].max() < QDOT_TOL and (gen["status"] == ref["status"]).all()
        gen_cold = bt.tick(d, DT, want_working_set=True)
        assert gen["iters"][ok].mean() <= gen_cold["iters"][ok].mean() + 1e-9
        bt.set_option("sim3_kernel", 1)
        bt.set_option("presolve", 1)
        back = bt.tick(dict(d, working_set=gen_cold["working_set"]), DT)
        assert np.abs(back["qdot"] - ref["qdot"])[ok].max() < QDOT_TOL
        its["general-path set into the compact kernel"] = back["iters"][ok].mean()
        same = (gen_cold["working_set"][ok] == cold["working_set"][ok]).all(axis=1).mean()
        assert same > 0.95, same                                           # both paths name the active constraints alike
    if cfg_name == "everything":
        assert bt.stat("last_path") == 3                               # working sets in and out on the packed orth kernel's INEQ variant
        bt.set_option("packed_orth", 0)                                # ... and the general kernel names the same constraints
        gen_cold = bt.tick(d, DT, want_working_set=True)
        assert bt.stat("last_path") == 0
        gen = bt.tick(dict(d, working_set=cold["working_set"]), DT, want_working_set=True)
        assert np.abs(gen["qdot"] - ref["qdot"])[ok].max() < QDOT_TOL and (gen["status"] == ref["status"]).all()
        bt.set_option("packed_orth", 2)
        back = bt.tick(dict(d, working_set=gen_cold["working_set"]), DT, want_working_set=True)
        assert bt.stat("last_path") == 3 and np.abs(back["qdot"] - ref["qdot"])[ok].max() < QDOT_TOL
        its["general-path set into the packed kernel"] = back["iters"][ok].mean()
        lock = np.where(mid == 0, (7 << 23) | (7 << (32 + 23)), (7 << 22) | (7 << (32 + 22)))     # (the general path also names the DoF the box locks at 0)
        same = (((gen_cold["working_set"][ok][:, 0] & ~lock[ok]) == (cold["working_set"][ok][:, 0] & ~lock[ok])) &
                ((gen_cold["working_set"][ok][:, 1] & 0x3F0000003F) == (cold["working_set"][ok][:, 1] & 0x3F0000003F))).mean()
        print("   packed and general kernel name the same active set on %.4f of the instances" % same)
        assert same > 0.95, same
    if cfg_name == "c2":
        bt.set_option("packed_orth", 2)                                # (B = 2048: the default policy keeps small batches on the one-instance kernel)
        got = bt.tick(dict(d, working_set=junk), DT, want_working_set=True)
        assert bt.stat("last_path") == 3                               # working sets do not send configs[1] off the packed orth kernel:
        assert (got["working_set"] == 0).all()                         # no inequality to seed, an empty set out
        assert (got["status"] == ref["status"]).all() and np.abs(got["qdot"] - ref["qdot"])[ok].max() < QDOT_TOL
    if cfg_name == "full":
        assert bt.stat("last_path") == 4                               # a working set was passed: the packed box kernel's WARM variant
        bt.set_option("packed_box", 0)                                 # ... and the general kernel names the same bounds
        gen = bt.tick(dict(d, working_set=cold["working_set"]), DT, want_working_set=True)
        assert bt.stat("last_path") == 0
        assert np.abs(gen["qdot"] - ref["qdot"])[ok].max() < QDOT_TOL and (gen["status"] == ref["status"]).all()
        gen_cold = bt.tick(d, DT, want_working_set=True)
        bt.set_option("packed_box", 1)
        back = bt.tick(dict(d, working_set=gen_cold["working_set"]), DT, want_working_set=True)
        assert bt.stat("last_path") == 4 and np.abs(back["qdot"] - ref["qdot"])[ok].max() < QDOT_TOL
        its["general-path set into the box kernel"] = back["iters"][ok].mean()
        # (the general path also names the DoF the box locks at 0 — equality bounds; the packed kernel leaves them out of its problem)
        lock = np.where(mid == 0, (7 << 23) | (7 << (32 + 23)), (7 << 22) | (7 << (32 + 22)))
        same = ((gen_cold["working_set"][ok][:, 0] & ~lock[ok]) == (cold["working_set"][ok][:, 0] & ~lock[ok])).mean()
        assert same > 0.95, same
    print(cfg_name, "active inequalities per instance %.2f; working-set changes per tick:" % nact[ok].mean(),
          ", ".join("%s %.2f" % kv for kv in its.items()))
    assert its["own"] <= its["cold"] + 1e-9 and its["previous tick"] <= its["cold"] + 0.25
    bt.close()
