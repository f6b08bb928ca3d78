#!/usr/bin/env python3
"""Headless replay of the reference's static-reach experiment (wrappers/sim3.py) on the MI355X path — SURVEY.md §8 f4.

sim3.py builds RobotModel, lets its constructor warm the robot up into the crouched stance, switches to
``setTasks(Grip=True, Joint="HYBRID")`` / ``setConstraints(Trunk, FR, FL, RR, RL)`` / ``staticReachMode()``
(sim3.py:145-148, 197), then walks the gripper target along a piecewise-linear milestone trajectory (klampt
``Trajectory.eval``; sim3.py:207-228: one milestone per unit of the parameter, advanced by 0.002 per tick = 500 ticks per
segment) while PyBullet plays the plant, and logs target-vs-reached gripper positions (sim3.py:340-348).

Here there is no PyBullet: the plant is the controller's own kinematic model (the commanded joint angles are reached
exactly, the IMU reports a level trunk), which is what ``wbc_rollout`` chains on the device — per tick wbc_tick →
wbc_update_state → reference-state side effects → target step. The warm-up and the set-up calls go through the
``RobotModel`` mirror (B = 1); the trajectory then runs for ``--batch`` instances at once (copies of the same robot whose
targets are shifted by a per-instance offset), one ``wbc_rollout`` call per trajectory segment.

    python tools/replay_sim3.py [--robot a1_wx200] [--batch 1] [--segments 3] [--ticks 500] [--posture HYBRID] [--csv out.csv]

The CSV has the columns of the reference's log (Time, target x/y/z, real x/y/z) for instance 0.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mech5845m-wbc-for-legged-manipulator_amd"))

# gripper milestones after the current position (data of sim3.py:211-212)
MILESTONES = {
    "a1_wx200": [[0.402, 0., 0.724], [0.402, 0.25, 0.724], [0.3, 0.35, 0.724], [0, 0.35, 0.724], [-0.1, 0.35, 0.48],
                 [-0.1, 0.35, 0.43], [0, 0.35, 0.24], [0.3, 0.35, 0.24], [0.402, 0.2, 0.24], [0.402, -0.2, 0.24],
                 [0.3, -0.35, 0.24], [0, -0.35, 0.24], [-0.1, -0.35, 0.43], [-0.1, -0.35, 0.48], [0, -0.35, 0.725],
                 [0.3, -0.35, 0.725], [0.402, -0.2, 0.725], [0.402, 0, 0.725]],
    "a1_px100_pin_ver": [[0.28, 0., 0.55], [0.28, -0.1, 0.55], [0.22, -0.2, 0.55], [0.07, -0.2, 0.55], [0., -0.2, 0.52],
                         [0., -0.2, 0.47], [0.07, -0.2, 0.44], [0.22, -0.20, 0.44], [0.32, -0.05, 0.44], [0.32, 0.1, 0.44],
                         [0.22, 0.2, 0.44], [0.07, 0.2, 0.44], [0, 0.2, 0.47], [0, 0.2, 0.52], [0.07, 0.2, 0.55],
                         [0.22, 0.2, 0.55], [0.28, 0.1, 0.55], [0.28, 0, 0.55]],
}


def build_robot(robot, posture):
    """sim3.py:60, 145-148, 197, 269 on the mirror."""
    import wbc_model
    from Robot_Wrapper4 import RobotModel
    r = wbc_model.A1_ROLES
    rm = RobotModel("/replay/%s.urdf" % robot, "/unused/meshes", r["EE_frame_names"], r["EE_joint_names"], r["G_base"],
                    r["imu"], "FR_hip_joint", r["hip_waist_joint_names"], foot_offset=True)
    rm.setTasks(Grip=True, Joint=posture)
    rm.setConstraints(Trunk=True, FR=True, FL=True, RR=True, RL=True)
    rm.staticReachMode()
    rm.initialiseWBC(np.array([0.0, 0.0, 0.0, 1.0]))
    return rm


def replay(rm, batch=1, segments=3, ticks=500, offsets=None, want_trace=True, robot="a1_wx200", sim3_kernel=1):
    """Walk the first `segments` segments of the milestone trajectory for `batch` instances; returns
    dict(time, target [K,3], real [K,3], status [batch], iters [batch], q [batch,27], seconds)."""
    from wbc_batch import WbcBatch
    model = rm._model
    cfg = rm._config()
    bt = WbcBatch(model, batch)
    bt.configure(cfg)
    bt.set_option("sim3_kernel", int(sim3_kernel))      # 0: the kernel RobotModel.runWBC's ticks run on (it passes orientation references)
    start = np.asarray(rm.prev_EE_pos[4], dtype=float).reshape(3)            # sim3.py:203: the trajectory starts where the gripper is
    pts = [start] + [np.array(m, dtype=float) for m in MILESTONES[robot]]
    EE_target = [np.asarray(rm.prev_EE_pos[i], dtype=float).reshape(3, 1) for i in range(5)]
    d1 = rm._tick_inputs(EE_target, None)
    d = {k: np.repeat(v, batch, axis=0) for k, v in d1.items()}
    if offsets is not None:                                                  # per-instance shift of the whole gripper trajectory
        d["ee_target"][:, 4] += offsets
        d["prev_ee_target"][:, 4] += offsets
    imu = np.tile(np.array([0.0, 0.0, 0.0, 1.0]), (batch, 1))
    targets, reals = [], []
    status = np.zeros(batch, dtype=np.int32)
    iters = np.zeros(batch, dtype=np.int64)
    t0 = time.perf_counter()
    for s in range(min(segments, len(pts) - 1)):
        step = np.zeros((batch, 5, 3))
        step[:, 4] = (pts[s + 1] - pts[s]) / ticks
        # the target of tick k is eval(k * 0.002): the first tick of a segment sits ON the milestone, like klampt's eval
        first_target = d["ee_target"][0, 4].copy()
        traces = []
        # The orientation references only matter while R*_prev != R* (the very first tick: prev_EE_CoM_rot comes from the
        # FK, R* from the stored Euler angles); afterwards qpb() has set R*_prev = R* and omega_ref = 0 exactly, which is what
        # passing no references means — and what lets the batch run on the compact sim3 kernel.
        chunks = [1, ticks - 1] if (s == 0 and "ee_ref_rot" in d and ticks > 1) else [ticks]
        for n in chunks:
            out = bt.rollout(d, rm.step_time, n, ee_target_step=step, imu=imu, want_trace=want_trace)
            status = np.maximum(status, out["status"])
            iters += out["iters"]
            if want_trace:
                traces.append(out["grip_trace"][:, 0])
            if len(chunks) == 2 and n == 1:
                d["q"] = out["q"]
                d["prev_ee_target"] = d["prev_ee_target"].copy()
                d["prev_ee_target"][:, 4] = d["ee_target"][:, 4]
                d["ee_target"] = out["ee_target"]
                d = {k: v for k, v in d.items() if k not in ("ee_ref_rot", "ee_prev_rot")}
        if want_trace:
            targets.append(first_target[None] + step[0, 4][None] * np.arange(ticks)[:, None])
            reals.append(np.concatenate(traces))
        # state for the next segment: what the reference's attributes hold after the last tick
        last_target = out["ee_target"] - step
        d["q"] = out["q"]
        d["prev_ee_target"] = d["prev_ee_target"].copy()
        d["prev_ee_target"][:, 4] = last_target[:, 4]                        # prev_EE_pos[4] = last target (RW4:1151)
        d["ee_target"] = out["ee_target"]
    seconds = time.perf_counter() - t0
    bt.close()
    res = dict(status=status, iters=iters, q=d["q"], seconds=seconds, ticks=ticks * min(segments, len(pts) - 1))
    if want_trace:
        res["target"], res["real"] = np.concatenate(targets), np.concatenate(reals)
        res["time"] = np.arange(len(res["target"])) * rm.step_time
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--robot", default="a1_wx200", choices=sorted(MILESTONES))
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--segments", type=int, default=3)
    ap.add_argument("--ticks", type=int, default=500, help="ticks per milestone (sim3.py: 1 / 0.002 = 500)")
    ap.add_argument("--posture", default="HYBRID")
    ap.add_argument("--csv", default=None)
    args = ap.parse_args()
    rm = build_robot(args.robot, args.posture)
    offsets = None
    if args.batch > 1:
        offsets = np.random.default_rng(0).uniform(-0.01, 0.01, (args.batch, 3))
        offsets[0] = 0
    out = replay(rm, args.batch, args.segments, args.ticks, offsets, robot=args.robot)
    err = np.linalg.norm(out["target"] - out["real"], axis=1)
    print("replayed %d ticks x %d instances in %.2f s (%.0f closed-loop ticks/s incl. host set-up per segment); "
          "worst status %d; gripper tracking error of instance 0: mean %.4f m, max %.4f m" % (
              out["ticks"], args.batch, out["seconds"], out["ticks"] * args.batch / out["seconds"], int(out["status"].max()),
              err.mean(), err.max()))
    if args.csv:
        with open(args.csv, "w") as f:
            f.write("Time,target x,target y,target z,real x,real y,real z\n")       # the columns of sim3.py:340-348
            for t, a, b in zip(out["time"], out["target"], out["real"]):
                f.write("%.3f,%.6f,%.6f,%.6f,%.6f,%.6f,%.6f\n" % (t, a[0], a[1], a[2], b[0], b[1], b[2]))


if __name__ == "__main__":
    main()
