#!/usr/bin/env python3
"""Latency of ONE stand-alone QP (the mirror's QP.solveQP / solveQPHotstart case: B = 1) at the tick's shape (m, n, p) = (32, 26, 16), on the packed
kernel and on the one-per-wavefront kernel (option packed_kernel 0): host buffers (the mirror's path) and device-resident buffers.
python tools/latency_qp_b1.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mech5845m-wbc-for-legged-manipulator_amd")]
import numpy as np, torch
from wbc_batch import WbcBatch
rng = np.random.default_rng(0)
m, n, p = 32, 26, 16
A = rng.normal(size=(1, m, n)); b = rng.normal(size=(1, m)); C = rng.normal(size=(1, p, n)); lb = -np.ones((1, n)) * 0.5; ub = -lb; cl = -np.ones((1, p)); cu = -cl
host = (A, b, C, lb, ub, cl, cu)
dev = [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in host]
bt = WbcBatch([], 1)
for pk in (1, 0):
    bt.set_option("packed_kernel", pk)
    for what, d in (("host buffers", host), ("device buffers", dev)):
        ts = []
        for i in range(300):
            t = time.perf_counter(); r = bt.qp_solve_ls(*d); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        ts = np.array(ts[50:]) * 1e6
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if d is dev:
            e0.record()
            for _ in range(200): bt.qp_solve_ls(*d)
            e1.record(); torch.cuda.synchronize()
            extra = "  device time %.1f us per call (HIP events over 200 calls)" % (e0.elapsed_time(e1) / 200 * 1e3)
        else:
            extra = ""
        print("QP(A, b) B = 1 (32, 26, 16), %s, %s: median %.1f us  p95 %.1f us  (iters %d, path %d)%s" % ("packed kernel" if pk else "one-per-wavefront kernel", what, np.median(ts), np.quantile(ts, 0.95), int(r[2][0]), bt.stat("last_qp_path"), extra), flush=True)
bt.close()
