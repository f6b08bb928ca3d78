#!/usr/bin/env python3
"""Throughput of the other BASELINE / test configurations (device-resident inputs, HIP events), for DESIGN.md's table.
The headline metric is bench.py's (config 3); these lines are context, not bench lines."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "mech5845m-wbc-for-legged-manipulator_amd")]
import numpy as np, torch, common, wbc_model
from wbc_batch import WbcBatch
wx, px = common.models()
dev = torch.device("cuda", 0)
only = sys.argv[1:]          # optional: configuration names to time (default: all)
for name, B, mixed in (("c2", 1024, False), ("c2", 4096, False), ("c2", 65536, False), ("c3", 1024, False), ("c3", 4096, False), ("c3", 65536, False),
                       ("c3", 65536, True), ("c3_trunk_task", 65536, False), ("c3_mani", 65536, False), ("everything", 65536, False), ("full", 1024, False), ("full", 4096, False), ("full", 65536, False),
                       ("hybrid_grip_com", 65536, False)):
    if only and name not in only:
        continue
    models = [wx, px] if mixed else [wx]
    cfgs = [common.config(name, m) for m in models]
    bt = WbcBatch(models, B)
    for i, c in enumerate(cfgs):
        bt.configure(c, i)
    if mixed:
        mid = (np.arange(B) % 2).astype(np.int32)
        parts = [common.tick_inputs(m, c, B, 5 + k) for k, (m, c) in enumerate(zip(models, cfgs))]
        d = {k: np.where(mid.reshape((B,) + (1,) * (parts[0][k].ndim - 1)) == 0, parts[0][k], parts[1][k]) for k in parts[0]}
        d["model_id"] = mid
    else:
        d = common.tick_inputs(wx, cfgs[0], B, 5, with_rot=(name in ("full", "everything")))
    dd = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in d.items()}
    out = dict(qdot=torch.zeros((B, 26), dtype=torch.float64, device=dev), status=torch.zeros(B, dtype=torch.int32, device=dev),
               iters=torch.zeros(B, dtype=torch.int32, device=dev))
    step = bt.make_tick_call(dd, out, 0.002)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%-16s B=%6d mixed=%d  m=%2d p=%2d  %.3f ms/step  %.1f M ticks/s  (optimal %.3f, iters %.1f, kernel path %d)" % (
        name, B, mixed, bt.task_rows, bt.constraint_rows, ms, B / ms / 1e3, (out["status"] == 0).double().mean().item(), out["iters"].double().mean().item(),
        bt.stat("last_path")))
    bt.close()
