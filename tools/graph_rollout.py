#!/usr/bin/env python3
"""Small-batch roll-outs: launch-bound. The same wbc_rollout call enqueued directly vs captured once into a HIP graph
(torch.cuda.CUDAGraph on the caller's stream) and replayed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "mech5845m-wbc-for-legged-manipulator_amd")]
import numpy as np, torch, common, wbc_model
from wbc_batch import WbcBatch
wx = wbc_model.load_model("a1_wx200")
cfg = common.config("c3", wx)
dev = torch.device("cuda", 0)
for B in (64, 1024, 8192):
    K = 50
    d = common.tick_inputs(wx, cfg, B, seed=3)
    dd = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in d.items()}
    step = torch.zeros((B, 5, 3), dtype=torch.float64, device=dev); step[:, 4, 0] = 1e-4
    bt = WbcBatch(wx, B); bt.configure(cfg)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        bt.rollout(dd, 0.002, K, ee_target_step=step, want_trace=False)
        side.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = bt.rollout(dd, 0.002, K, ee_target_step=step, want_trace=False)
        side.synchronize()
        t_direct = (time.perf_counter() - t0) / 5
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out_g = bt.rollout(dd, 0.002, K, ee_target_step=step, want_trace=False)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / 5
    same = torch.equal(out["q"], out_g["q"])
    print("B=%5d K=%d: direct %.3f ms (%.1f us/tick), graph %.3f ms (%.1f us/tick), identical results: %s" % (
        B, K, 1e3 * t_direct, 1e6 * t_direct / K, 1e3 * t_graph, 1e6 * t_graph / K, same))
    bt.close()
