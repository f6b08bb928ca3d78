"""CPU: the oracle's closed-loop pieces — updateState(running=True) with trunkWorldPos restated line by
line in numpy (reference wrappers/Robot_Wrapper4.py:387-428, 1297-1327), and the K-tick roll-out's behaviour."""
import numpy as np
import pytest

import common
import oracle
import wbc_capi as capi
import wbc_model
import wbc_workload

DT = 0.002


@pytest.fixture(scope="module")
def wx():
    return wbc_model.load_model("a1_wx200")


def test_update_state_restated_in_numpy(wx):
    rng = np.random.default_rng(12)
    B = 6
    q_cur = wbc_workload.sample_q(wx, B, rng)
    q_next = wbc_workload.sample_q(wx, B, rng)
    imu = rng.normal(size=(B, 4))
    imu /= np.linalg.norm(imu, axis=1, keepdims=True)
    targets = rng.normal(size=(B, 5, 3))
    got = oracle.update_state([wx], q_cur, q_next, targets, imu)
    got_noimu = oracle.update_state([wx], q_cur, q_next, targets, None)
    for b in range(B):
        for im, res in ((imu[b], got[b]), (q_next[b, 3:7], got_noimu[b])):
            config = np.concatenate((q_cur[b, :3], im, q_next[b, 7:]))                 # :388-389
            oMf = oracle.fk([wx], config[None], want_com=False)["oMf"][0]
            WRB = oMf[capi.FR_TRUNK, :9].reshape(3, 3)                                 # :1298
            trunk_pos = oMf[capi.FR_TRUNK, 9:].reshape(3, 1)                           # :1300
            FR, FL, RR, RL = (targets[b, i].reshape(3, 1) for i in range(4))
            BPAs = [oMf[capi.FR_EE0 + i, 9:].reshape(3, 1) - trunk_pos for i in range(4)]
            WPA = (FR + FL + RR + RL) / 4                                              # :1321
            BPA = (BPAs[0] + BPAs[1] + BPAs[2] + BPAs[3]) / 4                          # :1323
            base = (WPA - np.dot(WRB, BPA)).reshape(3,)                                # :1325
            want = np.concatenate((base, config[3:]))                                  # :415
            assert np.abs(res - want).max() < 1e-14


def level_inputs(model, cfg, B, seed, monkeypatch):
    """tick inputs with a level base (the robot of sim3.py stands on flat ground): the base estimator rotates a WORLD-frame
    offset by the trunk rotation once more (:1302-1325), which is only consistent for a level trunk."""
    orig = wbc_workload.sample_q

    def level(m, n, rng):
        q = orig(m, n, rng)
        q[:, 3:7] = [0.0, 0.0, 0.0, 1.0]
        return q
    monkeypatch.setattr(wbc_workload, "sample_q", level)
    return common.tick_inputs(model, cfg, B, seed=seed, stress=False)


def test_rollout_approaches_a_static_target(wx, monkeypatch):
    """Closed loop, static-reach preset: the gripper target sits 2 cm from the gripper; the reached position closes a good part of
    the gap within 80 ticks (Kp = 0.05 per tick, Robot_Wrapper4.py:1070, :1433; velocity bounds slow the first ticks).
    No physical claim beyond that: the "contact" rows are WORLD-frame Jacobian rows, so the stance feet
    are not actually pinned and the base estimator keeps re-anchoring them — that is the reference's behaviour."""
    B = 4
    cfg = common.config("c3", wx)
    d = level_inputs(wx, cfg, B, 5, monkeypatch)
    oMf0 = oracle.fk([wx], d["q"], want_com=False)["oMf"]
    goal = oMf0[:, capi.FR_EE0 + 4, 9:] + np.array([0.015, 0.005, 0.01])
    d["ee_target"][:, 4] = goal
    d["prev_ee_target"][:, 4] = goal
    imu = np.tile(d["q"][:, 3:7], 1)
    out = oracle.rollout([wx], [cfg], d, DT, B, 80, imu=imu, nthreads=4)
    assert (out["status"] == 0).all()
    err = np.linalg.norm(out["grip_trace"] - goal[None], axis=2)          # [K, B]
    assert (err.min(axis=0) < 0.6 * err[0]).all(), (err[0], err.min(axis=0))
    # the base quaternion is the IMU's, every tick; targets did not move and the reference state followed them
    assert (out["q"][:, 3:7] == imu).all()
    assert (out["ee_target"] == d["ee_target"]).all() and (out["prev_ee_target"][:, 4] == goal).all()
    # the foot targets were not tasks (only the gripper is, sim3.py:145): their prev entries keep their seeds
    assert (out["prev_ee_target"][:, :4] == d["prev_ee_target"][:, :4]).all()


def test_rollout_follows_a_moving_target(wx, monkeypatch):
    """One linear segment of sim3.py's milestone trajectory (sim3.py:207-228): the target moves 0.1 mm per tick; the
    feed-forward term (x* - x*_prev)/dt (Robot_Wrapper4.py:1063) makes the gripper keep pace."""
    B = 2
    cfg = common.config("c3", wx)
    d = level_inputs(wx, cfg, B, 6, monkeypatch)
    oMf0 = oracle.fk([wx], d["q"], want_com=False)["oMf"]
    start = oMf0[:, capi.FR_EE0 + 4, 9:].copy()
    d["ee_target"][:, 4] = start
    d["prev_ee_target"][:, 4] = start
    step = np.zeros((B, 5, 3))
    step[:, 4] = [1e-4, 0.0, 5e-5]
    K = 60
    out = oracle.rollout([wx], [cfg], d, DT, B, K, ee_target_step=step, nthreads=4)
    assert (out["status"] == 0).all()
    assert np.abs(out["ee_target"][:, 4] - (start + K * step[:, 4])).max() < 1e-12
    # weak closed-loop check only (the controller's tracking quality is the reference's business): the gripper travelled
    # along the segment's direction
    travel = (out["grip_trace"][-1] - start) @ (step[0, 4] / np.linalg.norm(step[0, 4]))
    assert (travel > 0.3 * np.linalg.norm((K - 1) * step[0, 4])).all(), travel


def test_warmup_reaches_the_crouched_stance_and_matches_its_fixture():
    """oracle.warmup = setInitialState (Robot_Wrapper4.py:196-351) for B robots: from the clamped neutral pose the feet end
    under their hips at 0.9 of their height and the gripper at its milestone (the straight-line trajectories are followed),
    the base quaternion's x, y, z are reset, the trunk height is -mean(foot z) + foot radius; and the committed fixture
    (tests/golden/warmup_mixed.npz, written by make_golden.py) is reproduced."""
    import os
    import make_golden
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "warmup_mixed.npz"))
    models = list(common.models())
    mid = z["model_id"]
    assert np.array_equal(make_golden.warmup_q0(models, mid, 13), z["q0"])
    sel = [0, 1, 2]                                  # the reference's own start pose of both robots + one perturbed
    out = oracle.warmup(models, z["q0"][sel], DT, 1000, foot_radius=float(z["foot_radius"]), model_id=mid[sel], nthreads=4)
    assert (out["status"] == 0).all()
    assert np.abs(out["q"] - z["out_q"][sel]).max() < 1e-9
    q = out["q"]
    assert (q[:, 3:6] == 0).all() and (q[:, 6] > 0.99).all()
    f = oracle.fk(models, q, mid[sel], want_com=False)["oMf"]
    feet, hips = f[:, capi.FR_EE0:capi.FR_EE0 + 4, 9:], f[:, capi.FR_HIP0:capi.FR_HIP0 + 4, 9:]
    assert np.abs(feet[:2, :, 2] - 0.02).max() < 2e-3                    # from the symmetric start pose the feet stand level on z = foot radius
    # (:336-337 SETS z to -mean(foot z) + radius, where foot z still contains the base height the warm-up drifted to (~0.2 mm):
    #  the mean foot height ends that much above the radius — the reference's arithmetic, reproduced)
    assert np.abs(feet[:, :, 2].mean(axis=1) - 0.02).max() < 1e-3
    assert np.abs(feet[:2, :, 0] - hips[:2, :, 0]).max() < 5e-3          # under their hips
    assert (q[:, 9] < -1.2).all() and (q[:, 8] > 0.5).all()              # knees bent: the crouch
    assert 0.25 < q[0, 2] < 0.33
    # milestones as the reference builds them (:238-262)
    assert np.allclose(out["goal"][:, :4, 2], 0.9 * out["start"][:, :4, 2]) and np.allclose(out["goal"][:, 4, 1], out["start"][:, 4, 1])
