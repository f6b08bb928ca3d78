"""CPU: the oracle's task/constraint assembly against independent numpy restatements of the reference's formulas
(wrappers/Robot_Wrapper4.py, line numbers in each test) and against the committed golden fixtures."""
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation as R

import common
import oracle
import wbc_capi as capi
import wbc_model

DT = 0.002
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def wx():
    return wbc_model.load_model("a1_wx200")


def test_row_counts_match_the_reference(wx):
    """sim3 tick: m = 6 + 26 = 32, p = 4 + 12 = 16; warm-up m = 36 + 26 = 62; config 2 m = 30 + 3 + 26 = 59."""
    assert (oracle.task_rows(common.config("c1", wx)), oracle.constraint_rows(common.config("c1", wx))) == (32, 16)
    assert (oracle.task_rows(common.config("full", wx)), oracle.constraint_rows(common.config("full", wx))) == (62, 0)
    assert (oracle.task_rows(common.config("c2", wx)), oracle.constraint_rows(common.config("c2", wx))) == (59, 12)


def test_task_rows_restated_in_numpy(wx):
    """endEffectorA2 :474-484, trunkA :487-490, qpJointA/b :1199-1268, calcTargetVelEE3 :1052-1157, calcTargetVelTrunk2 :948-1015."""
    cfg = common.config("everything", wx)
    B = 6
    d = common.tick_inputs(wx, cfg, B, seed=9, with_rot=True)
    a = oracle.assemble([wx], [cfg], d, DT, B)
    fk = oracle.fk([wx], d["q"])
    for b in range(B):
        q = d["q"][b]
        row = 0
        for e in range(5):
            J = oracle.frame_jacobian(wx, q, frame=e, rf=2)
            w = cfg.ee_w[e]
            assert np.abs(a["A"][b, row:row + 6] - J * w).max() < 1e-13
            xt, xp, x = d["ee_target"][b, e], d["prev_ee_target"][b, e], fk["oMf"][b, e, 9:]
            v = (xt - xp) / DT + cfg.ee_gain[e][0] * (xt - x) / DT
            Rs, Rp = d["ee_ref_rot"][b, e].reshape(3, 3), d["ee_prev_rot"][b, e].reshape(3, 3)
            S = ((Rs - Rp) / DT) @ Rs.T
            want = np.concatenate([v, [S[2, 1], S[0, 2], S[1, 0]]]) * w
            assert np.abs(a["b"][b, row:row + 6] - want).max() < 1e-9
            row += 6
        Jt = oracle.frame_jacobian(wx, q, frame=capi.FR_TRUNK, rf=0)
        assert np.abs(a["A"][b, row:row + 6] - Jt).max() < 1e-13
        xt, xp, x = d["trunk_target"][b], d["prev_trunk_target"][b], fk["oMf"][b, capi.FR_TRUNK, 9:]
        v = (xt - xp) / DT + cfg.trunk_gain[0] * (xt - x) / DT
        fq = R.from_matrix(fk["oMf"][b, capi.FR_TRUNK, :9].reshape(3, 3)).as_quat()
        ref = R.from_euler("xyz", d["trunk_ref_euler"][b])
        rq, Rs = ref.as_quat(), ref.as_matrix()
        qe = np.array([fq[3] * rq[0] - fq[0] * rq[3] + fq[1] * rq[2] - fq[2] * rq[1],
                       fq[3] * rq[1] - fq[1] * rq[3] - fq[0] * rq[2] + fq[2] * rq[0],
                       fq[0] * rq[1] - fq[1] * rq[0]])                       # :976 with its self-cancelling pair removed
        S = ((Rs - d["trunk_prev_rot"][b].reshape(3, 3)) / DT) @ Rs          # :984 (not transposed)
        want = np.concatenate([v, np.array([S[2, 1], S[0, 2], S[1, 0]]) + cfg.trunk_gain[3] * qe])
        assert np.abs(a["b"][b, row:row + 6] - want).max() < 1e-9
        row += 6
        assert np.abs(a["A"][b, row:row + 3] - fk["Jcom"][b]).max() < 1e-15        # Robot_Wrapper2.py:600-603
        want = d["com_target_vel"][b] + (d["com_target"][b] - fk["com"][b])         # Robot_Wrapper2.py:661-668
        assert np.abs(a["b"][b, row:row + 3] - want).max() < 1e-12
        row += 3
        dd = cfg.joint_w / 26
        assert np.abs(a["A"][b, row:row + 26] - np.eye(26) * dd).max() < 1e-18
        assert np.abs(a["b"][b, row:row + 26] - dd * np.delete(q, 6)).max() < 1e-15  # "PREV": :1216-1217
        # QP_Wrapper.py:17-18
        assert np.abs(a["H"][b] - a["A"][b].T @ a["A"][b]).max() < 1e-10
        assert np.abs(a["g"][b] + a["A"][b].T @ a["b"][b]).max() < 1e-9


def test_constraint_rows_restated_in_numpy(wx):
    """CoMConstraint :669-694, trunkConstraint :707-754, EEConstraint :757-761, order of findConstraints :764-836."""
    cfg = common.config("everything", wx)
    B = 4
    d = common.tick_inputs(wx, cfg, B, seed=10)
    a = oracle.assemble([wx], [cfg], d, DT, B)
    fk = oracle.fk([wx], d["q"])
    for b in range(B):
        q = d["q"][b]
        com, fl, rr = fk["com"][b], fk["oMf"][b, 1, 9:], fk["oMf"][b, 2, 9:]
        assert np.abs(a["C"][b, :2] - fk["Jcom"][b, :2]).max() < 1e-15
        assert np.abs(a["Clb"][b, :2] - (rr[:2] - com[:2]) / DT * 0.8).max() < 1e-10
        assert np.abs(a["Cub"][b, :2] - (fl[:2] - com[:2]) / DT * 0.8).max() < 1e-10
        Jt = oracle.frame_jacobian(wx, q, frame=capi.FR_TRUNK, rf=2)
        assert np.abs(a["C"][b, 2:6] - Jt[2:]).max() < 1e-15
        Mt = fk["oMf"][b, capi.FR_TRUNK]
        cur = np.concatenate([[Mt[11]], R.from_matrix(Mt[:9].reshape(3, 3)).as_euler("xyz")])
        bc = d["trunk_box_center"][b]
        var = np.array([bc[0] * 0.25, 0.15, 0.15, 0.15])
        assert np.abs(a["Clb"][b, 2:6] - 0.5 * ((bc - var) - cur) / DT).max() < 1e-9
        assert np.abs(a["Cub"][b, 2:6] - 0.5 * ((bc + var) - cur) / DT).max() < 1e-9
        for e in range(4):
            Jw = oracle.frame_jacobian(wx, q, frame=e, rf=0)
            assert np.abs(a["C"][b, 6 + 3 * e:9 + 3 * e] - Jw[:3]).max() < 1e-15          # WORLD frame (quirk C.2)
        assert np.abs(a["Clb"][b, 6:]).max() == 0 and np.abs(a["Cub"][b, 6:]).max() == 0


def _damper_literal(model, q, compat):
    """velDamperJointConstraints :572-637 restated literally in Python (index quirk included when compat)."""
    nv = model.nv
    lo, hi, vm = model.q_lo.copy(), model.q_hi.copy(), model.v_max.copy()
    grip = model.ee_joint[4]
    for i in range(len(lo)):
        if i < 7:
            lo[i], hi[i] = -5, 5
            if compat or i < 6:
                vm[i] = 5
        if i >= grip - 2 + 7:
            lo[i] = hi[i] = 0
    lo, hi = np.delete(lo, 6), np.delete(hi, 6)
    lb, ub = np.zeros(nv), np.zeros(nv)
    for i in range(nv):
        c = q[i] if compat else (q[i] if i < 6 else q[i + 1])
        if c <= lo[i] + 0.026:
            lb[i] = min(max(-0.01 * (c - lo[i] - 0.015) / (0.026 - 0.015), -vm[i]), vm[i])
        else:
            lb[i] = -vm[i]
        if c >= hi[i] - 0.026:
            ub[i] = min(max(0.01 * (hi[i] - c - 0.015) / (0.026 - 0.015), -vm[i]), vm[i])
        else:
            ub[i] = vm[i]
    lb, ub = np.where(lb > 0, -lb, lb), np.where(ub < 0, -ub, ub)
    lb[grip - 2 + 6:] = 0
    ub[grip - 2 + 6:] = 0
    return lb, ub


@pytest.mark.parametrize("compat", [True, False])
def test_velocity_damper_bounds_both_index_maps(wx, compat):
    cfg = wbc_model.make_config(wx, Grip=True, Joint="PREV", damper_compat=compat)
    B = 64
    d = common.tick_inputs(wx, cfg, B, seed=12)
    rng = np.random.default_rng(0)
    for b in range(B):       # push entries near limits so the damper zone is exercised under either map
        i = int(rng.integers(7, 24))
        d["q"][b, i] = (wx.q_lo[i] + rng.uniform(0, 0.03)) if rng.random() < 0.5 else (wx.q_hi[i] - rng.uniform(0, 0.03))
    a = oracle.assemble([wx], [cfg], d, DT, B)
    hit = 0
    for b in range(B):
        lb, ub = _damper_literal(wx, d["q"][b], compat)
        assert np.abs(a["lb"][b] - lb).max() < 1e-13 and np.abs(a["ub"][b] - ub).max() < 1e-13
        hit += int((np.abs(lb[6:23]) < 1).any() or (np.abs(ub[6:23]) < 1).any())
    assert hit > 10
    assert (a["lb"][:, 23:] == 0).all() and (a["ub"][:, 23:] == 0).all()      # locked gripper + fingers (:627-630)
    if compat:
        assert a["ub"][0, 6] <= 5.0       # vel_lim[6] = 5 quirk (:590-593): FL_hip's 52.4 rad/s is overwritten


GOLDEN_CFG = {"tick_c1": "c1", "tick_c2": "c2", "tick_c3": "c3", "tick_c5_mixed": "c3", "tick_everything": "everything",
              "tick_c3_hybrid": "c3_hybrid", "tick_c3_mani": "c3_mani"}


@pytest.mark.parametrize("name", sorted(GOLDEN_CFG))
def test_oracle_reproduces_golden_fixtures(name):
    """Regression anchor (tests/golden/README.md): the oracle's outputs on the committed seeded inputs."""
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    wx, px = common.models()
    d = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    B = d["q"].shape[0]
    mixed = "model_id" in d
    models = [wx, px] if mixed else [wx]
    cfgs = [common.config(GOLDEN_CFG[name], m) for m in models]
    out = oracle.tick(models, cfgs, d, DT, B)
    assert (out["status"] == z["out_status"]).all()
    assert np.abs(out["qdot"] - z["out_qdot"]).max() < 1e-9
    if not mixed:
        a = oracle.assemble(models, cfgs, d, DT, B)
        for k in ("H", "g", "C", "lb", "ub"):
            assert np.abs(a[k] - z["asm_" + k]).max() < 1e-12 * max(1, np.abs(z["asm_" + k]).max())


@pytest.mark.parametrize("mode", ["MANI", "HYBRID"])
def test_posture_target_restated_in_numpy(wx, mode):
    """qpJointb "MANI" (Robot_Wrapper4.py:1220-1242) and "HYBRID" (:1245-1260) replayed line by line in Python on top of
    the oracle's getJointJacobian, numpy's det included — the C restatement must give the same u and leave the same q."""
    import wbc_workload
    rng = np.random.default_rng(8)
    cfg = wbc_model.sim3_config(wx, Joint=mode)
    nv, arm_base_id = wx.nv, wx.joint_id("waist")
    assert cfg.arm_base_id == arm_base_id == 14
    for q0 in wbc_workload.sample_q(wx, 3, rng):
        def f(qq, joint_id):
            J = oracle.frame_jacobian(wx, qq, joint=joint_id, rf=2)[:, :nv]
            return np.sqrt(np.linalg.det(J @ J.T))
        q = q0.copy()
        deltaq = 0.0002
        if mode == "MANI":
            u = []
            for i in range(nv):
                joint_id = 1 if i < 6 else i + 1 - 5
                q[i] = q[i] + deltaq
                f1 = f(q, joint_id)
                q[i] = q[i] - (deltaq * 2)
                f2 = f(q, joint_id)
                u.append(0.5 * (f1 - f2) / deltaq)
            u = np.array(u)
        else:
            u = np.delete(q0, 6)
            for i in range(len(u)):
                joint_id = i - 6
                if joint_id >= arm_base_id:
                    q[i] = q[i] + deltaq
                    f1 = f(q, joint_id)
                    q[i] = q[i] - (deltaq * 2)
                    f2 = f(q, joint_id)
                    u[i] = 0.5 * (f1 - f2) / deltaq
        uo, qa = oracle.posture_target([wx], [cfg], q0[None])
        assert np.abs(uo[0] - u).max() < 1e-9
        assert (qa[0] == q).all()
        # the posture rows of b carry (1/nv) u w, and the rest of the tick runs at the perturbed q
        d = common.tick_inputs(wx, cfg, 1, seed=3)
        d["q"] = q0[None].copy()
        a = oracle.assemble([wx], [cfg], d, DT, 1)
        assert np.abs(a["b"][0, 6:] - (1.0 / nv) * u * cfg.joint_w).max() < 1e-12
        cfg_prev = wbc_model.sim3_config(wx, Joint="PREV")
        d2 = dict(d)
        d2["q"] = q[None].copy()
        a2 = oracle.assemble([wx], [cfg_prev], d2, DT, 1)
        for k in ("C", "Clb", "Cub", "lb", "ub"):
            assert (a[k] == a2[k]).all(), k


def test_hybrid_posture_is_prev_plus_zero_gradient(wx):
    """What HYBRID evaluates to with the reference's indices: DoF 20..25 differentiate joints 14..19 with respect to
    the angle of the joint AFTER them, on which a joint's own Jacobian cannot depend -> exactly 0; the rest is PREV."""
    import wbc_workload
    q = wbc_workload.sample_q(wx, 16, np.random.default_rng(1))
    u, qa = oracle.posture_target([wx], [wbc_model.sim3_config(wx, Joint="HYBRID")], q)
    assert (u[:, :20] == np.delete(q, 6, axis=1)[:, :20]).all()
    assert np.abs(u[:, 20:]).max() < 1e-9
    assert np.allclose((qa - q)[:, 20:26], -0.0002, rtol=1e-9, atol=0) and (qa[:, :20] == q[:, :20]).all()


@pytest.mark.parametrize("name,cfg_name", [("rollout_c3", "c3"), ("rollout_c3_hybrid", "c3_hybrid")])
def test_oracle_reproduces_rollout_fixtures(name, cfg_name):
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    wx, _ = common.models()
    cfg = common.config(cfg_name, wx)
    d = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    out = oracle.rollout([wx], [cfg], d, DT, d["q"].shape[0], int(z["ticks"]), ee_target_step=z["step"], imu=z["imu"])
    assert (out["status"] == z["out_status"]).all() and (out["iters"] == z["out_iters"]).all()
    assert np.abs(out["q"] - z["out_q"]).max() < 1e-10 and np.abs(out["grip_trace"] - z["out_grip_trace"]).max() < 1e-10
