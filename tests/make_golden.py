#!/usr/bin/env python3
"""Regenerate tests/golden/tick_*.npz: seeded tick inputs + the CPU oracle's outputs for BASELINE configs 1/2/3/5.
Runs the oracle only (no reference, no GPU). The reference itself cannot be executed here (pinocchio/qpOASES absent),
so these are the oracle's own regression anchors, not reference outputs (tests/golden/README.md)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import conftest  # noqa: F401,E402  (sets sys.path)
import common  # noqa: E402
import oracle  # noqa: E402

DT = 0.002


def make(name, cfg_name, B, seed, mixed=False):
    wx, px = common.models()
    if mixed:
        models = [wx, px]
        cfgs = [common.config(cfg_name, m) for m in models]
        mid = (np.arange(B) % 2).astype(np.int32)
        parts = [common.tick_inputs(m, c, B, seed + i) for i, (m, c) in enumerate(zip(models, cfgs))]
        d = {k: np.where(mid.reshape((B,) + (1,) * (parts[0][k].ndim - 1)) == 0, parts[0][k], parts[1][k]) for k in parts[0]}
        d["model_id"] = mid
    else:
        models, cfgs = [wx], [common.config(cfg_name, wx)]
        d = common.tick_inputs(wx, cfgs[0], B, seed, with_rot=(cfg_name in ("full", "everything")))
    out = oracle.tick(models, cfgs, d, DT, B)
    asm = oracle.assemble(models, cfgs, d, DT, B) if not mixed else {}
    np.savez_compressed(os.path.join(HERE, "golden", name + ".npz"), **{"in_" + k: v for k, v in d.items()},
                        **{"out_" + k: v for k, v in out.items()}, **{"asm_" + k: v for k, v in asm.items()})
    print(name, "status", np.bincount(out["status"]), "iters", out["iters"].tolist())


def make_rollout(name, cfg_name, B, seed, K):
    """K closed-loop ticks: inputs, per-tick steps and the oracle's final state / gripper trace."""
    wx, _ = common.models()
    cfg = common.config(cfg_name, wx)
    d = common.tick_inputs(wx, cfg, B, seed)
    rng = np.random.default_rng(seed)
    step = np.zeros((B, 5, 3))
    step[:, 4] = rng.normal(0, 1e-4, (B, 3))
    imu = d["q"][:, 3:7].copy()
    out = oracle.rollout([wx], [cfg], d, DT, B, K, ee_target_step=step, imu=imu)
    np.savez_compressed(os.path.join(HERE, "golden", name + ".npz"), **{"in_" + k: v for k, v in d.items()},
                        step=step, imu=imu, ticks=K, **{"out_" + k: v for k, v in out.items()})
    print(name, "status", np.bincount(out["status"]), "iters", out["iters"].tolist())


def make_kinematics():
    """(q -> oMf, data.J, the five LWA end-effector Jacobians, com, Jcom) for the poses SURVEY.md §8c lists: neutral,
    Robot_Wrapper.py:28's stand_joint_config (numbers only), 8 rows of the mocap gait log, 8 random in-limit q."""
    import wbc_workload
    wx, _ = common.models()
    rng = np.random.default_rng(11)
    stand = np.array([0., 0., 0., -0.001, 0.012, -0.001, 1., 0.007, 0.868, -1.168, -0.001, 0.869, -1.169, 0.007, 0.823, -1.094,
                      -0.001, 0.823, -1.094, -0.002, -2.133, 0.945, 1.112, 0.002, 0., 0., 0.])
    legs = wbc_workload.mocap_legs()[::32][:8].reshape(8, 4, 3)[:, [1, 0, 3, 2], :].reshape(8, 12)
    gait = np.tile(wx.neutral(), (8, 1))
    gait[:, 2] = 0.3
    gait[:, 7:19] = legs
    gait[:, 25], gait[:, 26] = 0.02, -0.02
    q = np.concatenate([wx.neutral()[None], stand[None], gait, wbc_workload.sample_q(wx, 8, rng)])
    out = oracle.fk([wx], q)
    Jee = np.array([[oracle.frame_jacobian(wx, qq, frame=e, rf=2) for e in range(5)] for qq in q])
    np.savez_compressed(os.path.join(HERE, "golden", "kinematics_wx200.npz"), q=q, oMf=out["oMf"], J=out["J"], com=out["com"],
                        Jcom=out["Jcom"], Jee_lwa=Jee)
    print("kinematics_wx200", q.shape)


def make_qp_cases():
    """Adversarial QPs at the QP(A, b, ...) boundary: every bound active, duplicated (consistent) equality
    rows, contradictory equality rows, an infeasible box, plus the unconstrained and bounds-only cases. Stored with the
    oracle's solution, status and the active set read off the solution."""
    rng = np.random.default_rng(21)
    n, p = 26, 8
    cases = {}

    def base():
        A = rng.normal(size=(40, n))
        return A.T @ A + 1e-3 * np.eye(n), rng.normal(size=n) * 3, rng.normal(size=(p, n))
    H, g, C = base()
    cases["unconstrained"] = (H, g, C, -np.full(n, 1e30), np.full(n, 1e30), -np.full(p, 1e30), np.full(p, 1e30))
    H, g, C = base()
    cases["bounds_only"] = (H, g, C, -np.full(n, 0.05), np.full(n, 0.05), -np.full(p, 1e30), np.full(p, 1e30))
    H, g, C = base()
    cases["all_bounds_active"] = (H, 1e3 * np.sign(rng.normal(size=n)), C, -np.full(n, 0.1), np.full(n, 0.1), -np.full(p, 1e30), np.full(p, 1e30))
    H, g, C = base()
    C[4:] = C[:4]                                    # duplicated rows, same right-hand sides: dependent but consistent
    rhs = rng.normal(size=4) * 0.1
    cl = np.concatenate([rhs, rhs])
    cases["duplicate_equalities"] = (H, g, C, -np.ones(n), np.ones(n), cl, cl.copy())
    cl2 = np.concatenate([rhs, rhs + 0.5])           # same rows, different right-hand sides: infeasible
    cases["contradictory_equalities"] = (H, g, C, -np.ones(n), np.ones(n), cl2, cl2.copy())
    H, g, C = base()
    C[0] = 0
    C[0, 3] = 1.0                                    # row 0 asks x3 >= 2 while the bound says x3 <= 1
    cl = -np.full(p, 1e30)
    cl[0] = 2.0
    cases["infeasible_box"] = (H, g, C, -np.ones(n), np.ones(n), cl, np.full(p, 1e30))
    H, g, C = base()
    lb, ub = -np.ones(n), np.ones(n)
    lb[-3:] = ub[-3:] = 0.0                          # the locked DoF of the reference (RW4:627-630)
    cases["locked_and_mixed"] = (H, g, C, lb, ub, np.concatenate([np.zeros(4), -np.full(4, 0.2)]), np.concatenate([np.zeros(4), np.full(4, 0.2)]))
    names = sorted(cases)
    arr = {k: np.array([cases[nm][i] for nm in names]) for i, k in enumerate(("H", "g", "C", "lb", "ub", "Clb", "Cub"))}
    x, st, it = oracle.qp_solve(arr["H"], arr["g"], arr["C"], arr["lb"], arr["ub"], arr["Clb"], arr["Cub"])
    act = np.zeros((len(names), n + p), dtype=np.int8)          # -1 at lower, +1 at upper, 2 equality
    for i in range(len(names)):
        v = np.concatenate([x[i], arr["C"][i] @ x[i]])
        lo, hi = np.concatenate([arr["lb"][i], arr["Clb"][i]]), np.concatenate([arr["ub"][i], arr["Cub"][i]])
        act[i] = np.where(lo == hi, 2, np.where(np.abs(v - lo) < 1e-8 * np.maximum(1, np.abs(lo)), -1,
                                                np.where(np.abs(v - hi) < 1e-8 * np.maximum(1, np.abs(hi)), 1, 0)))
    np.savez_compressed(os.path.join(HERE, "golden", "qp_cases.npz"), names=np.array(names), x=x, status=st, iters=it,
                        active=act, **arr)
    print("qp_cases", dict(zip(names, st.tolist())), it.tolist())


def warmup_q0(models, mid, seed):
    """start configurations for the warm-up: the reference's own (pin.neutral with every entry above its upper limit clamped,
    Robot_Wrapper4.py:199-208 — that bends the knees: the calf's upper limit is negative) for instance 0 of each model,
    perturbed arm / leg angles for the others."""
    rng = np.random.default_rng(seed)
    q0 = np.zeros((len(mid), 27))
    for b, i in enumerate(mid):
        m = models[i]
        q = m.neutral()
        for k in range(m.nv):
            if q[k] > m.q_hi[k]:
                q[k] = m.q_hi[k]
        if b >= len(models):
            q[7:m.nq - 3] += rng.normal(0, 0.05, m.nq - 10)
        q0[b] = q
    return q0


def make_warmup():
    """setInitialState: start configurations and the oracle's warmed-up state, both morphologies."""
    models = list(common.models())
    mid = np.array([0, 1, 0, 1, 0, 1], dtype=np.int32)
    q0 = warmup_q0(models, mid, 13)
    out = oracle.warmup(models, q0, DT, 1000, foot_radius=0.02, model_id=mid, nthreads=4)
    np.savez_compressed(os.path.join(HERE, "golden", "warmup_mixed.npz"), q0=q0, model_id=mid, foot_radius=0.02, ticks_per_segment=1000,
                        **{"out_" + k: v for k, v in out.items()})
    print("warmup_mixed status", out["status"].tolist(), "iters", out["iters"].tolist())


if __name__ == "__main__":
    make_warmup()
    if len(sys.argv) > 1 and sys.argv[1] == "warmup":
        sys.exit(0)
    make_kinematics()
    make_qp_cases()
    make("tick_c3_hybrid", "c3_hybrid", 8, 7)
    make("tick_c3_mani", "c3_mani", 4, 8)
    make_rollout("rollout_c3", "c3", 8, 9, 6)
    make_rollout("rollout_c3_hybrid", "c3_hybrid", 4, 10, 4)
    make("tick_c1", "c1", 1, 1)
    make("tick_c2", "c2", 8, 2)
    make("tick_c3", "c3", 16, 3)
    make("tick_c5_mixed", "c3", 8, 5, mixed=True)
    make("tick_everything", "everything", 8, 6)
