#!/usr/bin/env python3
"""Stage ablation of the packed stand-alone QP kernel (csrc/wbc_k_qpp.hip) on the ablation build (make -C csrc ablate): option dbg_stop = 400 + k
returns after stage k — 1 loads, 2 H = A'A and g staged, 3 Cholesky / substitution, 4 equalities and x_eq, 5 one violation scan — and the differences
are the stages' shares of QP(A, b) at (m, n, p). python tools/ablate_qpp.py [m n p]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("WBC_HIP_LIB", os.path.join(ROOT, "mech5845m-wbc-for-legged-manipulator_amd", "csrc", "build", "libwbc_hip_ablate.so"))
sys.path[:0] = [os.path.join(ROOT, "mech5845m-wbc-for-legged-manipulator_amd")]
import numpy as np, torch, wbc_model
from wbc_batch import WbcBatch
m, n, p = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 26, 16)
B = 32768
rng = np.random.default_rng(0)
A = rng.normal(size=(B, m, n)); b = rng.normal(size=(B, m)); C = rng.normal(size=(B, p, n)); lb = -np.ones((B, n)) * 0.5; ub = -lb; cl = -np.ones((B, p)); cu = -cl
bt = WbcBatch(wbc_model.load_model("a1_wx200"), B)
d = [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (A, b, C, lb, ub, cl, cu)]
prev = 0.0
for cut, name in ((401, "loads"), (402, "H = A'A, g, staging"), (403, "Cholesky + substitution"), (404, "equalities, x_eq"), (405, "one violation scan"), (0, "dual iterations (+ refinement)")):
    bt.set_option("dbg_stop", cut)
    for _ in range(3): bt.qp_solve_ls(*d)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): bt.qp_solve_ls(*d)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print("cut %3d  %-32s cumulative %7.1f us   stage %7.1f us" % (cut, name, dt * 1e6, (dt - prev) * 1e6), flush=True)
    prev = dt
bt.close()
