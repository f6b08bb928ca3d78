#!/usr/bin/env python3
"""Single-robot latency (B = 1) of the drop-in classes: what one sim3.py tick costs on the MI355X path, host buffers."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "mech5845m-wbc-for-legged-manipulator_amd")]
import numpy as np
import replay_sim3 as rp
rm = rp.build_robot("a1_wx200", "HYBRID")
imu = np.array([0.0, 0.0, 0.0, 1.0])
EE = [np.asarray(rm.prev_EE_pos[i], dtype=float).reshape(3, 1).copy() for i in range(5)]
for _ in range(20):
    rm.runWBC(imu, target_cartesian_pos_EE=EE, target_cartesian_pos_trunk=None)
ts = []
for k in range(200):
    EE[4] = EE[4] + np.array([[1e-4], [0.0], [5e-5]])
    t0 = time.perf_counter()
    rm.runWBC(imu, target_cartesian_pos_EE=EE, target_cartesian_pos_trunk=None)
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print("RobotModel.runWBC (B = 1, host buffers, HYBRID): median %.3f ms, p95 %.3f ms, max %.3f ms per tick" % (np.median(ts), np.quantile(ts, 0.95), ts.max()))
# the bare fused tick through WbcBatch
from wbc_batch import WbcBatch
bt = WbcBatch(rm._model, 1); bt.configure(rm._config())
d = rm._tick_inputs(EE, None)
for _ in range(20):
    bt.tick(d, 0.002, want_q_next=True)
ts = []
for _ in range(200):
    t0 = time.perf_counter(); bt.tick(d, 0.002, want_q_next=True); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print("WbcBatch.tick (B = 1, host buffers): median %.3f ms, p95 %.3f ms" % (np.median(ts), np.quantile(ts, 0.95)))
