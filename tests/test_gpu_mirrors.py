"""GPU: the reference-shaped Python classes (QP_Wrapper.QP, Robot_Wrapper4.RobotModel) at B = 1 against the oracle,
and the HIP path against the committed golden fixtures."""
import os

import numpy as np
import pytest

import common
import oracle
import wbc_capi as capi
import wbc_model

pytestmark = pytest.mark.gpu
DT = 0.002
HERE = os.path.dirname(os.path.abspath(__file__))


def test_qp_class_matches_reference_surface_and_oracle():
    from QP_Wrapper import QP
    wx = wbc_model.load_model("a1_wx200")
    cfg = common.config("c1", wx)
    d = common.tick_inputs(wx, cfg, 3, seed=50)
    a = oracle.assemble([wx], [cfg], d, DT, 3)
    ref = oracle.tick([wx], [cfg], d, DT, 3)
    A, b, C = a["A"][0], a["b"][0], a["C"][0]
    Cview = C.T                                        # what findConstraints returns (Robot_Wrapper4.py:836)
    qp = QP(A, b, a["lb"][0], a["ub"][0], Cview, a["Clb"][0], a["Cub"][0], n_of_velocity_dimensions=26)
    x = qp.solveQP()
    assert x is qp.xOpt and x.shape == (26,) and qp.status == 0
    assert np.abs(x - ref["qdot"][0]).max() < 1e-7      # (the refinement, QP_Wrapper.py:37 numRefinementSteps, reaches the mirror: the oracle refines too)
    assert np.abs(qp.H - A.T @ A).max() < 1e-12 and np.abs(qp.g + A.T @ b).max() < 1e-12      # QP_Wrapper.py:17-18
    for k in (1, 2):                                   # hotstart: new H, g, C each call; the same ndarray comes back
        y = qp.solveQPHotstart(a["A"][k], a["b"][k], a["lb"][k], a["ub"][k], a["C"][k].T, a["Clb"][k], a["Cub"][k])
        assert y is x
        assert np.abs(y - ref["qdot"][k]).max() < 1e-5 and qp.status == 0 and int(qp.nWSR[0]) > 0
    # it IS a hot start (qp.hotstart, QP_Wrapper.py:70): the same problem again is seeded with its own final working set — one step per
    # active inequality on top of the 12 contact equalities + 3 locked DoF, never more working-set changes than the cold solve
    k = 2
    y = qp.solveQPHotstart(a["A"][k], a["b"][k], a["lb"][k], a["ub"][k], a["C"][k].T, a["Clb"][k], a["Cub"][k])
    assert np.abs(y - ref["qdot"][k]).max() < 1e-5 and int(qp.nWSR[0]) <= int(ref["iters"][k])
    cold = QP(a["A"][k], a["b"][k], a["lb"][k], a["ub"][k], a["C"][k].T, a["Clb"][k], a["Cub"][k], n_of_velocity_dimensions=26)
    cold.solveQP()
    assert int(cold.nWSR[0]) == int(ref["iters"][k]) and (cold._ws == qp._ws).all()
    # bounds-only problem (QProblemB branch, QP_Wrapper.py:25-26) and the 2n-long bound vectors of Robot_Wrapper2
    qb = QP(A, b, np.concatenate([a["lb"][0], a["lb"][0]]), np.concatenate([a["ub"][0], a["ub"][0]]), n_of_velocity_dimensions=26)
    xb = qb.solveQP()
    xr, st, _ = oracle.qp_solve(A.T @ A, -A.T @ b, None, a["lb"][0], a["ub"][0])
    assert st == 0 and np.abs(xb - xr).max() < 1e-5
    qm = QP(A, b, a["lb"][0], a["ub"][0], Cview, a["Clb"][0], a["Cub"][0], n_of_velocity_dimensions=26)
    qm.use_mfma = True                                  # J'J on the fp64 matrix cores: same answer
    assert np.abs(qm.solveQP() - ref["qdot"][0]).max() < 1e-5
    # the refinement (QP_Wrapper.py:37 numRefinementSteps) reaches the mirror: QP(A, b, ...) is within 1e-7 of the oracle, which refines too
    assert np.abs(qm.xOpt - ref["qdot"][0]).max() < 1e-7
    # the reference's state machine: a cold solveQP carries no working set — neither into it nor past a FAILED one. An infeasible
    # problem (contradictory trunk box) through solveQP, then a feasible hot start: solved from an empty set, like a fresh QProblem's first hotstart
    bad_lo, bad_hi = a["Clb"][0].copy(), a["Cub"][0].copy()
    bad_lo[0], bad_hi[0] = 50.0, 60.0                   # trunk z rate far beyond every velocity bound
    qf = QP(A, b, a["lb"][0], a["ub"][0], Cview, a["Clb"][0], a["Cub"][0], n_of_velocity_dimensions=26)
    qf.solveQP()
    assert qf.status == 0 and qf._ws is not None
    qf.Clb, qf.Cub = bad_lo, bad_hi
    xf = qf.solveQP()
    assert qf.status != 0 and qf._ws is None and (xf == 0).all()          # unsolved: zeros (the fresh xOpt of :50), nothing carried
    yh = qf.solveQPHotstart(a["A"][1], a["b"][1], a["lb"][1], a["ub"][1], a["C"][1].T, a["Clb"][1], a["Cub"][1])
    assert qf.status == 0 and np.abs(yh - ref["qdot"][1]).max() < 1e-5 and int(qf.nWSR[0]) == int(ref["iters"][1])   # a cold count: no seed


def test_integration_md_ctypes_stub_runs_as_written():
    """INTEGRATION.md §2 shows the ctypes binding a maintainer of the reference would add to QP_Wrapper.py. Execute that very block (only the
    library path is substituted): cold solve = the oracle's answer, hot-started solve = the same answer in no more working-set changes."""
    import re
    root = os.path.dirname(HERE)
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "wbc_qp_solve_ls.argtypes" in b)
    stub = stub.replace('C.CDLL("libwbc_hip.so")', "C.CDLL(%r)" % capi.LIB_PATH)
    ns = {}
    exec(compile(stub, "INTEGRATION.md#stub", "exec"), ns)
    wx = wbc_model.load_model("a1_wx200")
    cfg = common.config("c1", wx)
    d = common.tick_inputs(wx, cfg, 2, seed=50)
    a = oracle.assemble([wx], [cfg], d, DT, 2)
    ref = oracle.tick([wx], [cfg], d, DT, 2)
    for k in range(2):
        args = (a["A"][k], a["b"][k], a["lb"][k].copy(), a["ub"][k].copy(), a["C"][k].T, a["Clb"][k].copy(), a["Cub"][k].copy(), 26)
        x, st = ns["_solve"](*args)                      # cold (solveQP)
        assert st == 0 and np.abs(x - ref["qdot"][k]).max() < 1e-5
        x2, st2 = ns["_solve"](*args, hot=True)          # solveQPHotstart: seeded with the set the cold solve left in _ws
        assert st2 == 0 and np.abs(x2 - ref["qdot"][k]).max() < 1e-5
        assert ns["_ws"].shape == (2,)                   # (the set that came back may be empty: equalities and locked DoF are not part of it)


GOLDEN_CFG = {"tick_c1": "c1", "tick_c2": "c2", "tick_c3": "c3", "tick_c5_mixed": "c3", "tick_everything": "everything",
              "tick_c3_hybrid": "c3_hybrid", "tick_c3_mani": "c3_mani"}


@pytest.mark.parametrize("name,cfg_name", [("rollout_c3", "c3"), ("rollout_c3_hybrid", "c3_hybrid")])
def test_hip_rollout_reproduces_golden_fixtures(name, cfg_name):
    from wbc_batch import WbcBatch
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    wx, _ = common.models()
    d = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    B = d["q"].shape[0]
    bt = WbcBatch(wx, B)
    bt.configure(common.config(cfg_name, wx))
    out = bt.rollout(d, DT, int(z["ticks"]), ee_target_step=z["step"], imu=z["imu"])
    assert (out["status"] == z["out_status"]).all()
    assert np.abs(out["q"] - z["out_q"]).max() < 1e-6 and np.abs(out["grip_trace"] - z["out_grip_trace"]).max() < 1e-6
    assert np.abs(out["qdot"] - z["out_qdot"]).max() < 1e-4
    bt.close()


@pytest.mark.parametrize("name", sorted(GOLDEN_CFG))
def test_hip_path_reproduces_golden_fixtures(name):
    from wbc_batch import WbcBatch
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    wx, px = common.models()
    d = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    B = d["q"].shape[0]
    mixed = "model_id" in d
    models = [wx, px] if mixed else [wx]
    cfg_name = GOLDEN_CFG[name]
    bt = WbcBatch(models, B)
    for i, m in enumerate(models):
        bt.configure(common.config(cfg_name, m), i)
    out = bt.tick(d, DT)
    assert (out["status"] == z["out_status"]).all()
    assert np.abs(out["qdot"] - z["out_qdot"]).max() < 1e-5
    if not mixed:
        a = bt.assemble(d, DT)
        for k in ("H", "g", "C", "lb", "ub"):
            assert np.abs(a[k] - z["asm_" + k]).max() < (1e-9 if "mani" in name else 1e-11) * max(1, np.abs(z["asm_" + k]).max())
    bt.close()


@pytest.fixture(scope="module")
def robot():
    from Robot_Wrapper4 import RobotModel
    r = wbc_model.A1_ROLES
    return RobotModel("/any/where/a1_wx200.urdf", "/unused/meshes", r["EE_frame_names"], r["EE_joint_names"], r["G_base"],
                      r["imu"], "FR_hip_joint", r["hip_waist_joint_names"], foot_offset=True)


def test_robot_model_warm_up_reaches_the_crouched_stance(robot):
    """setInitialState (Robot_Wrapper4.py:196-351): 2000 bounds-only QPs drag the neutral pose to the stance."""
    rm = robot
    assert rm.initialised and rm.foot_radius == 0.02
    q = rm.current_joint_config
    assert q.shape == (27,) and abs(np.linalg.norm(q[3:7]) - 1) < 1e-6 and np.abs(q[3:6]).max() == 0
    feet_z = [rm.EE_frame_pos[i][2] for i in range(4)]
    assert max(feet_z) - min(feet_z) < 5e-3                      # all four feet on one plane
    # base z := -mean(foot z) + foot radius (:336-337) is computed with world-frame foot heights, so the feet end at
    # radius - (base drift during the warm-up): within a few cm of the ground plane, exactly as the reference would.
    assert abs(np.mean(feet_z) - 0.02) < 0.05
    assert 0.25 < q[2] < 0.40                                    # 0.9 x the neutral leg length
    assert (q[7:19] > rm.robot_model.lowerPositionLimit[7:19] - 1e-6).all() and (q[7:19] < rm.robot_model.upperPositionLimit[7:19] + 1e-6).all()
    assert len(rm.FL_leg) == 3 and len(rm.grip) == 8
    # the gripper went up and forward of its neutral position (multiplier_G, :240-241, :252-261)
    assert rm.EE_frame_pos[4][2] - q[2] > 0.15


def test_robot_model_warm_up_is_the_oracles_warm_up():
    """RobotModel.setInitialState is the B = 1 call of the batched device warm-up: the state a fresh object ends up in is the
    oracle's restatement of Robot_Wrapper4.py:196-351 from the same clamped neutral pose (not just a plausible stance)."""
    from Robot_Wrapper4 import RobotModel
    r = wbc_model.A1_ROLES
    rm = RobotModel("/any/where/a1_wx200.urdf", "/unused/meshes", r["EE_frame_names"], r["EE_joint_names"], r["G_base"],
                    r["imu"], "FR_hip_joint", r["hip_waist_joint_names"], foot_offset=True)
    wx = rm._model
    q0 = wx.neutral()
    for i in range(wx.nv):
        if q0[i] > wx.q_hi[i]:
            q0[i] = wx.q_hi[i]
    ref = oracle.warmup([wx], q0[None], 0.002, 1000, foot_radius=0.02, nthreads=4)
    assert ref["status"][0] == 0 and rm.solver_status == 0
    assert np.abs(rm.current_joint_config - ref["q"][0]).max() < 1e-6
    assert np.abs(np.array(rm.FL_leg) - ref["q"][0, 7:10]).max() < 1e-6
    # the reference state the loop leaves behind: targets at their last milestones
    assert np.abs(np.array(rm.FR_target_cartesian_pos).reshape(3) - ref["goal"][0, 0]).max() < 1e-12


def test_robot_model_tick_matches_oracle(robot):
    """sim3's call sequence (sim3.py:145-148, 197, 269, 314) on the mirror; every quantity re-derived by the oracle."""
    rm = robot
    rm.setTasks(Grip=True, Joint="PREV")
    rm.setConstraints(Trunk=True, FR=True, FL=True, RR=True, RL=True)
    rm.staticReachMode()
    imu = np.array([0.01, -0.02, 0.005, 1.0])
    imu /= np.linalg.norm(imu)
    rm.initialiseWBC(imu)
    wx = rm._model
    EE_target = [rm.prev_EE_pos[i].reshape(3, 1).copy() for i in range(5)]
    trunk_target = rm.robot_data.oMf[rm.trunk_frame_index].translation.reshape(3, 1).copy()
    for tick in range(3):
        EE_target[4] = EE_target[4] + np.array([[0.0005], [0.0002], [0.0004]])
        cfg = rm._config()
        d = rm._tick_inputs(EE_target, trunk_target)
        a = oracle.assemble([wx], [cfg], d, rm.step_time, 1)
        ref = oracle.tick([wx], [cfg], d, rm.step_time, 1)
        # accessors = what the reference's runWBC gathers (:1348-1361)
        A = rm.qpA()
        Ct, Clb, Cub = rm.findConstraints()
        lb, ub = rm.velDamperJointConstraints()
        assert A.shape == (32, 26) and Ct.shape == (26, 16)
        assert np.abs(A - a["A"][0]).max() < 1e-12 and np.abs(Ct.T - a["C"][0]).max() < 1e-12
        assert np.abs(Clb - a["Clb"][0]).max() < 1e-9 and np.abs(lb - a["lb"][0]).max() < 1e-12 and np.abs(ub - a["ub"][0]).max() < 1e-12
        base_before = np.array(rm.current_joint_config[:3])
        FL, FR, RL, RR, grip = rm.runWBC(imu, target_cartesian_pos_EE=EE_target, target_cartesian_pos_trunk=trunk_target)
        assert rm.solver_status == 0
        assert np.abs(rm.q_vel - ref["qdot"][0]).max() < 1e-5
        joints = ref["q_next"][0, 7:]
        assert np.abs(np.concatenate([FL, FR, RL, RR, grip]) - joints).max() < 1e-7
        assert (len(FL), len(FR), len(RL), len(RR), len(grip)) == (3, 3, 3, 3, 8)
        # updateState(running=True): base xyz re-estimated from the stance feet (:414-415, :1297-1327), quaternion = IMU
        q = rm.current_joint_config
        assert np.abs(q[3:7] - imu).max() == 0 and np.abs(q[7:] - joints).max() < 1e-7
        # trunkWorldPos restated with the oracle's FK at [old base xyz, IMU quaternion, new joints]: WPA - WRB @ BPA with
        # BPA the mean WORLD-frame foot offset from the trunk (the reference rotates it once more; replicated literally)
        q_first = np.concatenate([base_before, imu, joints])
        o = oracle.fk([wx], q_first[None], want_com=False)["oMf"][0]
        Rt, pt = o[capi.FR_TRUNK, :9].reshape(3, 3), o[capi.FR_TRUNK, 9:]
        tgt = np.array([np.asarray(EE_target[i]).reshape(3) for i in range(4)])
        want_base = tgt.mean(0) - Rt @ (o[:4, 9:].mean(0) - pt)
        assert np.abs(q[:3] - want_base).max() < 1e-9
        assert np.abs(np.asarray(rm.prev_EE_pos[4]).reshape(3) - EE_target[4].reshape(3)).max() == 0
    # single-block accessors
    rm.endEffectorA2(4)
    Jg = oracle.frame_jacobian(wx, np.concatenate([rm.current_joint_config]), frame=4, rf=2)
    assert np.abs(rm.EE_A_list[4] - Jg).max() < 1e-12
    C, l, u = rm.EEConstraint(1)
    assert np.abs(C - oracle.frame_jacobian(wx, rm.current_joint_config, frame=1, rf=0)[:3]).max() < 1e-12 and not l.any() and not u.any()
    C, l, u = rm.CoMConstraint()
    assert np.abs(C - oracle.fk([wx], rm.current_joint_config[None])["Jcom"][0, :2]).max() < 1e-12


def test_robot_model_hybrid_posture_like_sim3(robot):
    """sim3.py:145 runs the controller with Joint="HYBRID": the posture target and the perturbed state qpJointb leaves
    behind come from the device; the oracle replays the same tick literally."""
    rm = robot
    rm.setTasks(Grip=True, Joint="HYBRID")
    rm.setConstraints(Trunk=True, FR=True, FL=True, RR=True, RL=True)
    rm.staticReachMode()
    imu = np.array([0.0, 0.0, 0.0, 1.0])
    rm.initialiseWBC(imu)
    wx = rm._model
    EE_target = [rm.prev_EE_pos[i].reshape(3, 1).copy() for i in range(5)]
    trunk_target = rm.robot_data.oMf[rm.trunk_frame_index].translation.reshape(3, 1).copy()
    for tick in range(2):
        EE_target[4] = EE_target[4] + np.array([[0.0004], [0.0], [0.0003]])
        cfg = rm._config()
        assert cfg.task_joint == capi.JOINT_HYBRID and cfg.posture_literal == 1
        d = rm._tick_inputs(EE_target, trunk_target)
        ref = oracle.tick([wx], [cfg], d, rm.step_time, 1)
        legs_grip = rm.runWBC(imu, target_cartesian_pos_EE=EE_target, target_cartesian_pos_trunk=trunk_target)
        assert rm.solver_status == 0 and ref["status"][0] == 0
        assert np.abs(rm.q_vel - ref["qdot"][0]).max() < 1e-5
        assert np.abs(np.concatenate(legs_grip) - ref["q_next"][0, 7:]).max() < 1e-7
        # the leak: the commanded arm joints carry the -2e-4 rad left by the finite differences, every tick
        plain = oracle.integrate([wx], d["q"], ref["qdot"], rm.step_time)[0, 7:]
        assert np.allclose((np.concatenate(legs_grip) - plain)[13:19], -0.0002, rtol=0, atol=1e-7)
    # the accessor sequence of runWBC (:1348-1361): qpb perturbs the state, findConstraints / dampers then read it
    EE_target[4] = EE_target[4] + np.array([[0.0004], [0.0], [0.0003]])
    cfg = rm._config()
    d = rm._tick_inputs(EE_target, trunk_target)
    a = oracle.assemble([wx], [cfg], d, rm.step_time, 1)
    q_before = rm.current_joint_config.copy()
    A = rm.qpA()
    b = rm.qpb(EE_target, trunk_target)
    Ct, Clb, Cub = rm.findConstraints()
    lb, ub = rm.velDamperJointConstraints()
    assert np.abs(A - a["A"][0]).max() < 1e-12 and np.abs(b.ravel() - a["b"][0]).max() < 1e-9
    assert np.allclose((rm.current_joint_config - q_before)[20:26], -0.0002, rtol=0, atol=1e-12)
    assert (rm.current_joint_config[:20] == q_before[:20]).all()
    assert np.abs(Ct.T - a["C"][0]).max() < 1e-12 and np.abs(Clb - a["Clb"][0]).max() < 1e-9
    assert np.abs(lb - a["lb"][0]).max() < 1e-12 and np.abs(ub - a["ub"][0]).max() < 1e-12
    rm.posture_literal = False
    q_before = rm.current_joint_config.copy()
    rm.qpb(EE_target, trunk_target)
    assert (rm.current_joint_config == q_before).all()
    rm.posture_literal = True


def test_headless_replay_equals_the_mirror_loop():
    """tools/replay_sim3.py (wbc_rollout per trajectory segment) against the same ticks issued one by one
    through RobotModel.runWBC, sim3.py's own loop (sim3.py:300-330) with the kinematic plant.
    The gripper trajectory is shifted 2 cm sideways: from the warmed-up, exactly left-right symmetric stance the reference's
    closed loop is unstable in the waist (q̇_waist grows 10^4-fold per tick from rounding noise until it chatters between its
    +-pi rad/s bounds; every tick still matches the oracle to < 3e-7: profiles/r02_replay_symmetric_stance.txt), so from
    there any two implementations that differ by one rounding end 1e-3 apart after 12 ticks."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("replay_sim3", os.path.join(os.path.dirname(HERE), "tools", "replay_sim3.py"))
    rp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rp)
    K = 12
    off = np.array([0.0, 0.02, 0.0])
    rm = rp.build_robot("a1_wx200", "HYBRID")
    out = rp.replay(rm, batch=3, segments=1, ticks=K, offsets=np.tile(off, (3, 1)))
    assert out["status"].max() == 0
    # the same K ticks on the mirror
    rm2 = rp.build_robot("a1_wx200", "HYBRID")
    start = np.asarray(rm2.prev_EE_pos[4], dtype=float).reshape(3)
    step = (np.array(rp.MILESTONES["a1_wx200"][0]) - start) / K
    EE_target = [np.asarray(rm2.prev_EE_pos[i], dtype=float).reshape(3, 1).copy() for i in range(5)]
    rm2.prev_EE_pos[4] = (start + off).reshape(3, 1)
    imu = np.array([0.0, 0.0, 0.0, 1.0])
    reals = []
    for k in range(K):
        EE_target[4] = (start + off + k * step).reshape(3, 1)
        rm2.runWBC(imu, target_cartesian_pos_EE=EE_target, target_cartesian_pos_trunk=None)
        assert rm2.solver_status == 0
        reals.append(rm2.robot_data.oMf[rm2.end_effector_index_list_frame[4]].translation.copy())
    # two different kernels (the roll-out runs on the compact sim3 kernel, runWBC passes orientation references and takes
    # the general one): q̇ agrees to ~1e-6 per tick along the weakly determined directions (cond(H) ~ 3e9), which the closed
    # loop carries forward
    assert np.abs(out["q"] - rm2.current_joint_config[None]).max() < 5e-6
    assert np.abs(out["real"] - np.array(reals)).max() < 5e-6
    assert np.abs(out["target"][-1] - (start + off + (K - 1) * step)).max() < 1e-12
    assert (out["q"][0] == out["q"][1]).all() and (out["q"][0] == out["q"][2]).all()      # identical instances stay identical


def test_hip_reproduces_kinematics_and_qp_fixtures():
    """The committed kinematics poses (neutral, stand, mocap, random) and the adversarial QPs through the C-ABI."""
    from wbc_batch import WbcBatch
    wx, _ = common.models()
    z = np.load(os.path.join(HERE, "golden", "kinematics_wx200.npz"))
    bt = WbcBatch(wx, 32)
    got = bt.fk(z["q"])
    for k in ("oMf", "J", "com", "Jcom"):
        assert np.abs(got[k] - z[k]).max() < 1e-12, k
    # the LWA end-effector Jacobians = the Grip/foot task rows with unit weights (endEffectorA2, RW4:474-484)
    cfg = wbc_model.make_config(wx, FR=True, FL=True, RR=True, RL=True, Grip=True, Joint=True)
    bt.configure(cfg)
    d = common.tick_inputs(wx, cfg, len(z["q"]), seed=0)
    d["q"] = z["q"]
    A = bt.assemble(d, DT, want=("A",))["A"]
    assert np.abs(A[:, :30].reshape(-1, 5, 6, 26) - z["Jee_lwa"]).max() < 1e-12
    zq = np.load(os.path.join(HERE, "golden", "qp_cases.npz"))
    x, st, it = bt.qp_solve(zq["H"], zq["g"], zq["C"], zq["lb"], zq["ub"], zq["Clb"], zq["Cub"])
    assert (st == zq["status"]).all(), dict(zip(zq["names"].tolist(), st.tolist()))
    ok = zq["status"] == 0
    assert np.abs(x - zq["x"])[ok].max() < 1e-9
    assert (it == zq["iters"])[ok].all()
    bt.close()


def test_qp_class_keeps_the_stale_answer_for_an_unsolved_qp():
    """qpOASES' getPrimalSolution does not write an unsolved problem's vector (QP_Wrapper.py:50-51, 71-73): zeros on the
    first QP, the previous answer on a hot start."""
    from QP_Wrapper import QP
    rng = np.random.default_rng(0)
    n, p = 6, 2
    A = rng.normal(size=(10, n))
    b = rng.normal(size=10)
    lb, ub = -np.ones(n), np.ones(n)
    C = np.zeros((n, p))
    C[0, 0] = 1.0
    C[1, 1] = 1.0
    qp = QP(A, b, lb, ub, C, np.array([-0.5, -0.5]), np.array([0.5, 0.5]), n_of_velocity_dimensions=n)
    x0 = qp.solveQP().copy()
    assert qp.status == 0 and np.abs(x0).max() > 0
    x1 = qp.solveQPHotstart(A, b, lb, ub, C, np.array([2.0, -0.5]), np.array([3.0, 0.5]))   # x0 >= 2 against the bound x0 <= 1
    assert qp.status == 2 and np.array_equal(x1, x0)
    bad = QP(A, b, lb, ub, C, np.array([2.0, -0.5]), np.array([3.0, 0.5]), n_of_velocity_dimensions=n)
    assert not bad.solveQP().any() and bad.status == 2
