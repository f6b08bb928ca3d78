"""CPU: pins the oracle's kinematics (oracle/wbc_oracle.c) — the reference's only numeric dump (Jacobians.py KATs),
finite differences, frame-convention identities, integrate round trips, scipy Rotation goldens."""
import json
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation as R

import common
import oracle
import wbc_capi as capi
import wbc_model
import wbc_workload

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def wx():
    return wbc_model.load_model("a1_wx200")


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(HERE, "golden", "jacobians_kat.json")) as f:
        return json.load(f)


def test_model_matches_survey_appendix_a(wx):
    assert (wx.nq, wx.nv, wx.njoints) == (27, 26, 22)
    assert wx.joint_names[4] == "FL_calf_joint" and wx.joint_names[19] == "gripper"   # Jacobians.py:1,18
    assert abs(wx.data["total_mass"] - 14.217857) < 1e-9
    px = wbc_model.load_model("a1_px100_pin_ver")
    assert (px.nq, px.nv, px.njoints) == (26, 25, 21) and px.joint_names[18] == "gripper"


def test_reference_joint_jacobian_kats(wx, kat):
    """tests_NOT_FOR_USE/Jacobians.py:1-24 — printed to 3-6 digits; col 20 row 0 of joint 19 is a sign typo there."""
    q = wx.neutral()
    for key, jid in (("joint19_gripper_world", 19), ("joint1_root_world", 1), ("joint4_FL_calf_world", 4)):
        want = np.array(kat[key])
        got = oracle.frame_jacobian(wx, q, joint=jid, rf=0)
        diff = np.abs(got - want)
        if jid == 19:
            assert abs(got[0, 20] + 0.362825) < 1e-12 and want[0, 20] == 0.362825   # documented typo
            diff[0, 20] = 0
        assert diff.max() < 5e-7, key


def test_reference_com_jacobian_structure(wx, kat):
    """Jacobians.py:27-41 is from an older inertial revision: zero pattern and signs agree, magnitudes within the
    inertial change (a few mm of CoM shift)."""
    want = np.array(kat["com_jacobian_neutral_old_inertials"])
    got = oracle.fk([wx], wx.neutral()[None])["Jcom"][0]
    assert np.abs(got[:, :3] - np.eye(3)).max() < 1e-15
    big = np.abs(want) > 5e-3          # entries proportional to the CoM height changed sign with the inertials (cz -2.2 mm -> +0.08 mm)
    assert big.sum() >= 20 and (np.sign(got[big]) == np.sign(want[big])).all()
    assert np.abs(got - want).max() < 4e-3
    com = oracle.fk([wx], wx.neutral()[None])["com"][0]
    assert np.abs(com - np.array([0.026805, 0.000720, 0.0000756])).max() < 1e-6


def _fd_jacobian(model, q0, fn, eps=1e-6):
    cols = []
    for k in range(model.nv):
        v = np.zeros((1, 26))
        v[0, k] = eps
        qp = oracle.integrate([model], q0[None], v, 1.0)
        qm = oracle.integrate([model], q0[None], -v, 1.0)
        cols.append((fn(qp) - fn(qm)) / (2 * eps))
    return np.array(cols).T


@pytest.mark.parametrize("name", ["a1_wx200", "a1_px100_pin_ver"])
def test_frame_and_com_jacobians_by_finite_differences(name):
    m = wbc_model.load_model(name)
    rng = np.random.default_rng(1)
    for q0 in wbc_workload.sample_q(m, 3, rng):
        for f in range(6):
            Jf = oracle.frame_jacobian(m, q0, frame=f, rf=2)
            Jfd = _fd_jacobian(m, q0, lambda q: oracle.fk([m], q, want_com=False)["oMf"][0, f, 9:])
            assert np.abs(Jfd - Jf[:3, :m.nv]).max() < 1e-8
        Jc = oracle.fk([m], q0[None])["Jcom"][0]
        Jfd = _fd_jacobian(m, q0, lambda q: oracle.fk([m], q)["com"][0])
        assert np.abs(Jfd - Jc[:, :m.nv]).max() < 1e-8


def test_reference_frame_identities(wx):
    """WORLD vs LOCAL_WORLD_ALIGNED vs LOCAL of getFrameJacobian."""
    rng = np.random.default_rng(2)
    q = wbc_workload.sample_q(wx, 1, rng)[0]
    oMf = oracle.fk([wx], q[None], want_com=False)["oMf"][0]
    for f in range(6):
        Rf, pf = oMf[f, :9].reshape(3, 3), oMf[f, 9:]
        JW, JL, JA = (oracle.frame_jacobian(wx, q, frame=f, rf=r) for r in (0, 1, 2))
        assert np.abs(JA[3:] - JW[3:]).max() == 0
        assert np.abs(JA[:3] - (JW[:3] + np.cross(JW[3:].T, pf).T)).max() < 1e-14
        assert np.abs(JL[:3] - Rf.T @ JA[:3]).max() < 1e-14 and np.abs(JL[3:] - Rf.T @ JA[3:]).max() < 1e-14
    # a foot frame is moved by the base and its own leg only (9 columns)
    J = oracle.frame_jacobian(wx, q, frame=0, rf=0)
    assert set(np.nonzero(np.abs(J).sum(0))[0]) == set(range(6)) | {9, 10, 11}      # FR leg = v 9..11


def test_integrate_round_trip_and_composition(wx):
    rng = np.random.default_rng(4)
    q = wbc_workload.sample_q(wx, 16, rng)
    v = rng.normal(size=(16, 26))
    qn = oracle.integrate([wx], q, v, 0.002)
    back = oracle.integrate([wx], qn, -v, 0.002)
    assert np.abs(back - q).max() < 1e-14
    assert np.abs(np.linalg.norm(qn[:, 3:7], axis=1) - 1).max() < 1e-15
    assert np.abs(qn[:, 7:] - (q[:, 7:] + 0.002 * v[:, 6:])).max() < 1e-16
    # pure rotation about base z by angle a: quaternion = q0 * (0, 0, sin a/2, cos a/2)
    w = np.zeros((1, 26))
    w[0, 5] = 0.7
    q1 = oracle.integrate([wx], q[:1], w, 1.0)[0]
    want = (R.from_quat(q[0, 3:7]) * R.from_rotvec([0, 0, 0.7])).as_quat()
    assert min(np.abs(q1[3:7] - want).max(), np.abs(q1[3:7] + want).max()) < 1e-14


def test_rotation_helpers_match_scipy():
    """scipy.spatial.transform.Rotation is what the reference calls (Robot_Wrapper4.py:222-226, 714-715, 964-970)."""
    quat_to_R, R_to_euler, euler_to_R, R_to_quat, euler_to_quat = oracle.rot_helpers()
    rng = np.random.default_rng(0)
    for _ in range(2000):
        e = rng.uniform(-np.pi, np.pi, 3)
        e[1] = rng.uniform(-1.5, 1.5)
        M = R.from_euler("xyz", e).as_matrix()
        assert np.abs(euler_to_R(e).reshape(3, 3) - M).max() < 1e-14
        assert np.abs(R_to_euler(M) - R.from_matrix(M).as_euler("xyz")).max() < 1e-12
        assert np.abs(R_to_quat(M) - R.from_matrix(M).as_quat()).max() < 1e-14
        assert np.abs(euler_to_quat(e) - R.from_euler("xyz", e).as_quat()).max() < 1e-14
        q = R.from_euler("xyz", e).as_quat()
        assert np.abs(quat_to_R(q).reshape(3, 3) - M).max() < 1e-14


def test_fk_against_hand_geometry(wx):
    """gripper_bar at neutral: x = 0.095+0.05+0.2+0.065+0.043, z = 0.058+0.066175+0.03865+0.2."""
    oMf = oracle.fk([wx], wx.neutral()[None], want_com=False)["oMf"][0]
    assert np.abs(oMf[capi.FR_EE0 + 4, 9:] - np.array([0.453, 0.0, 0.362825])).max() < 1e-15
    assert np.abs(oMf[capi.FR_EE0 + 0, 9:] - np.array([0.183, -0.13205, -0.4])).max() < 1e-15   # FR foot


def test_kinematics_fixture():
    """tests/golden/kinematics_wx200.npz: neutral, Robot_Wrapper.py:28's stand_joint_config, 8 mocap rows,
    8 random poses -> oMf, data.J, the five LWA end-effector Jacobians, com, Jcom."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kinematics_wx200.npz"))
    wx = wbc_model.load_model("a1_wx200")
    out = oracle.fk([wx], z["q"])
    for k in ("oMf", "J", "com", "Jcom"):
        assert np.abs(out[k] - z[k]).max() < 1e-13, k
    Jee = np.array([[oracle.frame_jacobian(wx, qq, frame=e, rf=2) for e in range(5)] for qq in z["q"]])
    assert np.abs(Jee - z["Jee_lwa"]).max() < 1e-13
    # sanity of the stored numbers themselves: at the stand pose the four feet are ~0.28 m below the trunk, level within 2 cm
    feet_z = z["oMf"][1, :4, 11]
    assert feet_z.max() - feet_z.min() < 0.02 and -0.33 < feet_z.mean() < -0.22
