#!/usr/bin/env python3
"""Throughput of the QP boundary on the TICK'S OWN problem — QP(A, b, lb, ub, C, Clb, Cub) as Robot_Wrapper4.runWBC hands it to QP_Wrapper (m = 32, n = 26,
p = 16 for the sim3 switch set: 12 contact equalities, 4 trunk-box rows, 26 bounds with 3 locked) — through wbc_qp_solve_ls (A, b) and wbc_qp_solve (H, g),
device-resident data from wbc_assemble, HIP events. python tools/time_qp_tick.py [B] [cfg]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "mech5845m-wbc-for-legged-manipulator_amd")]
import numpy as np, torch, common
from wbc_batch import WbcBatch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
name = sys.argv[2] if len(sys.argv) > 2 else "c3"
wx, _ = common.models()
cfg = common.config(name, wx)
bt = WbcBatch(wx, B); bt.configure(cfg)
d = common.tick_inputs(wx, cfg, B, 5)
dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items()}
a = bt.assemble(dev, 0.002)
m, p = a["A"].shape[1], a["C"].shape[1]
tick = bt.tick(dev, 0.002)


def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): r = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r


for rf in (0, 1):
    bt.set_option("refine", rf)
    for label, f in (("QP(A, b) VALU J'J", lambda: bt.qp_solve_ls(a["A"], a["b"], a["C"], a["lb"], a["ub"], a["Clb"], a["Cub"], use_mfma=False)),
                     ("QP(A, b) MFMA J'J", lambda: bt.qp_solve_ls(a["A"], a["b"], a["C"], a["lb"], a["ub"], a["Clb"], a["Cub"], use_mfma=True)),
                     ("QP(H, g)", lambda: bt.qp_solve(a["H"], a["g"], a["C"], a["lb"], a["ub"], a["Clb"], a["Cub"]))):
        ms, r = timeit(f)
        x, st, it = r[:3]
        ok = (st == 0) & (tick["status"] == 0)
        err = (x - tick["qdot"]).abs().max(dim=1).values[ok].max().item()
        print("%s (m, n, p) = (%d, 26, %d) B = %d refine %d  %-18s %.3f ms  %.1f M QPs/s  optimal %.3f  iters %.1f  |x - wbc_tick| %.1e" % (
            name, m, p, B, rf, label, ms, B / ms / 1e3, (st == 0).double().mean().item(), it.double().mean().item(), err), flush=True)
bt.set_option("refine", 1)
ms, _ = timeit(lambda: bt.tick(dev, 0.002))
print("%s the fused wbc_tick on the same instances (kinematics and assembly included): %.3f ms  %.1f M ticks/s" % (name, ms, B / ms / 1e3))
bt.close()
