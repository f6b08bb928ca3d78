#!/usr/bin/env python3
"""One timing case for the matrix-core evidence (tools/mfma_evidence.sh runs it under rocprofv3, once per case):
    python3 tools/time_mfma.py tick <config> <B> <mfma 0|1>      wbc_tick on the general kernel (sim3 kernel off)
    python3 tools/time_mfma.py qpls <m> <B> <mfma 0|1>           wbc_qp_solve_ls: H = A'A (m x 26) + QP (QP_Wrapper.py:17-18)
Prints one line: case, ms per launch, M instances/s (device-resident inputs, HIP events, 10 launches after 3 warm-ups)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "mech5845m-wbc-for-legged-manipulator_amd")]
import numpy as np, torch, common
from wbc_batch import WbcBatch

kind, what, B, mfma = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda", 0)
wx = common.models()[0]
if kind == "tick":
    cfg = common.config(what, wx)
    bt = WbcBatch(wx, B)
    bt.configure(cfg)
    bt.set_option("sim3_kernel", 0)          # same (general) kernel on both arms: only the contraction differs
    bt.set_option("presolve", 0)
    bt.set_option("jtj_mfma", mfma)
    d = common.tick_inputs(wx, cfg, B, 5, with_rot=(what in ("full", "everything")))
    dd = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in d.items()}
    out = dict(qdot=torch.zeros((B, 26), dtype=torch.float64, device=dev), status=torch.zeros(B, dtype=torch.int32, device=dev),
               iters=torch.zeros(B, dtype=torch.int32, device=dev))
    step = bt.make_tick_call(dd, out, 0.002)
    label = "tick %s m=%d p=%d" % (what, bt.task_rows, bt.constraint_rows)
else:
    m, n, p = int(what), 26, 16
    rng = np.random.default_rng(7)
    A = rng.normal(size=(B, m, n)) * 0.3
    A[:, m - n:, :] += np.eye(n)[None] * 0.5          # keeps H well conditioned
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    Ad, bd, Cd = t(A), t(rng.normal(size=(B, m))), t(rng.normal(size=(B, p, n)))
    lb, ub, cl, cu = t(-np.ones((B, n))), t(np.ones((B, n))), t(-0.3 * np.ones((B, p))), t(0.3 * np.ones((B, p)))
    bt = WbcBatch(wx, B)
    step = lambda: bt.qp_solve_ls(Ad, bd, Cd, lb, ub, cl, cu, use_mfma=bool(mfma))
    label = "qp_solve_ls m=%d n=%d p=%d" % (m, n, p)
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
print("MFMA_CASE %-28s B=%6d jtj=%s  %.3f ms/launch  %.2f M/s" % (label, B, "mfma_f64" if mfma else "valu_f64", ms, B / ms / 1e3), flush=True)
bt.close()
