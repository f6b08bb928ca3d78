#!/usr/bin/env python3
"""Step latency of BASELINE configs[1] at its own batch (B = 1024) on the packed orth kernel vs the one-instance kernel, for several disjoint
subsets of the seeded inputs (a subset with a flagged stance-leg block pays the tail's general-path tick on top).   python3 tools/orthp_latency.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mech5845m-wbc-for-legged-manipulator_amd")]
import numpy as np, torch
import wbc_model, wbc_workload
from wbc_batch import WbcBatch
wx = wbc_model.load_model("a1_wx200")
cfg = wbc_model.equality_only_config(wx)
B, NS = 1024, 8
bt = WbcBatch(wx, B * NS)
bt.configure(cfg)

class FK:
    def __call__(_, q):
        return bt.fk(q, want=("oMf",))["oMf"]
    def com(_, q):
        return bt.fk(q, want=("com",))["com"]
d = wbc_workload.make_tick_inputs(wx, cfg, B * NS, 5, FK())
for k in range(NS):
    dev = {n: torch.from_numpy(np.ascontiguousarray(v[k * B:(k + 1) * B])).cuda() for n, v in d.items()}
    out = dict(qdot=torch.zeros((B, 26), dtype=torch.float64, device="cuda"), status=torch.zeros(B, dtype=torch.int32, device="cuda"),
               iters=torch.zeros(B, dtype=torch.int32, device="cuda"))
    line = "subset %d:" % k
    for po in (2, 0):
        bt.set_option("packed_orth", po)
        step = bt.make_tick_call(dev, out, 0.002)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            step()
        e1.record()
        torch.cuda.synchronize()
        line += "  %s %.1f us (path %d, flagged %d)" % ("packed" if po else "one-instance", 1e3 * e0.elapsed_time(e1) / 50, bt.stat("last_path"), bt.stat("deferred_last"))
    print(line, flush=True)
bt.close()
