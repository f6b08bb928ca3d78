#!/usr/bin/env python3
"""Refinement check of the QP entry points (GPU box): QP(A, b, ...) = wbc_qp_solve_ls and QP(H, g) = wbc_qp_solve on the benchmark tick's own
data (the oracle's assembly of C3: 12 contact equalities + 3 fixed variables, cond(H) ~ 3e9) and on random ill-conditioned problems, refine 0 / 1,
against the oracle's refined answers. python tools/refine_check_qp.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "mech5845m-wbc-for-legged-manipulator_amd")]
import numpy as np, oracle, common
from wbc_batch import WbcBatch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
wx, px = common.models()
bt = WbcBatch([wx], B)
threads = min(32, len(os.sched_getaffinity(0)))


def report(tag, got, ref):
    x, st, it = got
    xr, sr, ir = ref
    ok = (sr == 0) & (st == 0)
    err = np.abs(x - xr).max(axis=1)[ok]
    print("%-58s status agree %.5f iters equal %.4f  err max %.2e p99 %.2e p50 %.2e" % (tag, (st == sr).mean(), (it == ir)[ok].mean(), err.max(), np.quantile(err, 0.99), np.median(err)), flush=True)


for cfg_name in ("c3", "c2", "everything"):
    cfg = common.config(cfg_name, wx)
    d = common.tick_inputs(wx, cfg, B, 3000, with_rot=(cfg_name == "everything"))
    a = oracle.assemble([wx], [cfg], d, 0.002, B)
    ref_ls = oracle.qp_solve_ls(a["A"], a["b"], a["C"], a["lb"], a["ub"], a["Clb"], a["Cub"], nthreads=threads)
    ref_h = oracle.qp_solve(a["H"], a["g"], a["C"], a["lb"], a["ub"], a["Clb"], a["Cub"], nthreads=threads)
    for rf in (0, 1):
        bt.set_option("refine", rf)
        for mf in (False, True):
            report("%s QP(A, b) refine %d mfma %d vs oracle ls" % (cfg_name, rf, mf), bt.qp_solve_ls(a["A"], a["b"], a["C"], a["lb"], a["ub"], a["Clb"], a["Cub"], use_mfma=mf), ref_ls)
        report("%s QP(H, g) refine %d vs oracle (H residual)" % (cfg_name, rf), bt.qp_solve(a["H"], a["g"], a["C"], a["lb"], a["ub"], a["Clb"], a["Cub"]), ref_h)
        report("%s QP(H, g) refine %d vs oracle ls" % (cfg_name, rf), bt.qp_solve(a["H"], a["g"], a["C"], a["lb"], a["ub"], a["Clb"], a["Cub"]), ref_ls)
rng = np.random.default_rng(5)
for (m, n, p) in ((18, 14, 6), (40, 26, 16), (12, 8, 0)):
    A = np.concatenate([rng.normal(size=(B, m - n, n)), np.broadcast_to(3e-5 * np.eye(n), (B, n, n))], axis=1)
    b = np.concatenate([rng.normal(size=(B, m - n)), 3e-5 * rng.normal(size=(B, n))], axis=1)
    C = rng.normal(size=(B, p, n)) if p else None
    lb, ub = -rng.uniform(0.5, 2.0, (B, n)), rng.uniform(0.5, 2.0, (B, n))
    cl, cu = (-rng.uniform(0.1, 1.0, (B, p)), rng.uniform(0.1, 1.0, (B, p))) if p else (None, None)
    if p:
        cl[:, 0] = cu[:, 0] = 0.05
    ref = oracle.qp_solve_ls(A, b, C, lb, ub, cl, cu, nthreads=threads)
    for rf in (0, 1):
        bt.set_option("refine", rf)
        report("random m %d n %d p %d QP(A, b) refine %d" % (m, n, p, rf), bt.qp_solve_ls(A, b, C, lb, ub, cl, cu), ref)
bt.close()
