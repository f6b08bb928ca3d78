#!/usr/bin/env python3
"""Step time of tests/common.py "everything" at small batches: the packed orth kernel's INEQ variant (option packed_orth = 2) against the one-instance
general kernel (0) — the batch-size policy of the packed orth kernel (from 4608 instances on) applies to both of its variants.
    python3 tools/small_batch_orthq.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mech5845m-wbc-for-legged-manipulator_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import common
from wbc_batch import WbcBatch
wx, px = common.models()
cfg = common.config("everything", wx)
bt = WbcBatch(wx, 8192)
bt.configure(cfg)
for B in (64, 256, 1024, 2048, 4096, 8192):
    d = common.tick_inputs(wx, cfg, B, 5, with_rot=True)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items()}
    out = dict(qdot=torch.zeros((B, 26), dtype=torch.float64, device="cuda"), status=torch.zeros(B, dtype=torch.int32, device="cuda"),
               iters=torch.zeros(B, dtype=torch.int32, device="cuda"))
    line = "B %5d:" % B
    for po in (2, 0):
        bt.set_option("packed_orth", po)
        step = bt.make_tick_call(dev, out, 0.002)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                step()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 50)
        line += "  %s %.1f us (path %d, redone in the tail %d)" % ("packed orth INEQ" if po else "one-instance", 1e3 * float(np.median(ts)), bt.stat("last_path"),
                                                               bt.stat("deferred_last") if po else 0)
    print(line, flush=True)
bt.close()
