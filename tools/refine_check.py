#!/usr/bin/env python3
"""Refinement check (GPU box): each configuration through the HIP path with option refine = 0 / 1 against the oracle (which refines); prints the
q̇ error distribution and the kernel path, and what one tick costs with and without the step. python tools/refine_check.py [B] [cfg ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "mech5845m-wbc-for-legged-manipulator_amd")]
import numpy as np, torch, oracle, common
from wbc_batch import WbcBatch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
names = sys.argv[2:] or ["c3", "c3_hybrid", "c3_trunk_task", "c3_mani", "c3_mixed", "c2", "full", "everything", "hybrid_grip_com", "c3_two_feet", "c3_nobounds"]
wx, px = common.models()
threads = min(32, len(os.sched_getaffinity(0)))
for name in names:
    mixed = name.endswith("_mixed")
    cfg_name = name[:-6] if mixed else name
    models = [wx, px] if mixed else [wx]
    cfgs = [common.config(cfg_name, m) for m in models]
    bt = WbcBatch(models, B)
    for i, c in enumerate(cfgs):
        bt.configure(c, i)
    if mixed:
        mid = (np.arange(B) % 2).astype(np.int32)
        parts = [common.tick_inputs(m, c, B, 2000 + 17 * k) for k, (m, c) in enumerate(zip(models, cfgs))]
        d = {k: np.where(mid.reshape((B,) + (1,) * (parts[0][k].ndim - 1)) == 0, parts[0][k], parts[1][k]) for k in parts[0]}
        d["model_id"] = mid
    else:
        d = common.tick_inputs(wx, cfgs[0], B, 2000, with_rot=(cfg_name in ("everything", "full")))
    ref = oracle.tick(models, cfgs, d, 0.002, B, nthreads=threads, want_q_next=False)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items()}
    for opts in ({"refine": 0}, {"refine": 1}, {"refine": 1, "packed_kernel": 0}, {"refine": 1, "packed_kernel": 0, "sim3_kernel": 0}, {"refine": 1, "presolve": 0}):
        for k, v in {"packed_kernel": 1, "sim3_kernel": 1, "presolve": 1}.items():
            bt.set_option(k, v)
        for k, v in opts.items():
            bt.set_option(k, v)
        got = bt.tick(dev, 0.002)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            got = bt.tick(dev, 0.002)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        st, qd = got["status"].cpu().numpy(), got["qdot"].cpu().numpy()
        ok = (ref["status"] == 0) & (st == 0)
        err = np.abs(qd - ref["qdot"]).max(axis=1)[ok]
        print("%-16s %-55s path %d orth %d: status agree %.6f, qdot err max %.2e p99.9 %.2e p50 %.2e, %.3f ms" % (
            name, str(opts), bt.stat("last_path"), bt.stat("last_orth"), (ref["status"] == st).mean(), err.max(), np.quantile(err, 0.999), np.median(err), ms), flush=True)
        if opts.get("refine") == 1 and len(opts) <= 2:      # the same tick hot-started with its own final working set (the WARM kernel variants)
            cold = bt.tick(dev, 0.002, want_working_set=True)
            warm = bt.tick(dict(dev, working_set=cold["working_set"]), 0.002)
            stw, qw = warm["status"].cpu().numpy(), warm["qdot"].cpu().numpy()
            okw = (ref["status"] == 0) & (stw == 0)
            ew = np.abs(qw - ref["qdot"]).max(axis=1)[okw]
            print("%-16s %-55s path %d       : status agree %.6f, qdot err max %.2e p99.9 %.2e p50 %.2e" % (
                name, "  hot-started with its own set", bt.stat("last_path"), (ref["status"] == stw).mean(), ew.max(), np.quantile(ew, 0.999), np.median(ew)), flush=True)
    bt.close()
