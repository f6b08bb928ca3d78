#!/usr/bin/env python3
"""What a rank-deficient stance-leg block costs: BASELINE workload (C3, B = 65536), share of flagged instances
set through option presolve_tol_exp (7 = default ~0 %, 5 ~ 1 %, 3 ~ 40 %, 0 = 100 % on all four legs), answered (a) by the pivoted
elimination inside the compact kernel, (b) by the second pass (dbg_force_defer: general kernel over the compact list); the general
kernel alone (presolve off) is the yardstick. ms per 65536-tick step, HIP events, device-resident inputs."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mech5845m-wbc-for-legged-manipulator_amd"))
import numpy as np, torch
import wbc_model, wbc_workload
from wbc_batch import WbcBatch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
model = wbc_model.load_model("a1_wx200")
cfg = wbc_model.sim3_config(model)
bt = WbcBatch(model, B)
bt.configure(cfg)
d = wbc_workload.make_tick_inputs(model, cfg, B, 0, lambda q: bt.fk(q, want=("oMf",))["oMf"])
dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items()}
out = dict(qdot=torch.zeros((B, 26), dtype=torch.float64, device="cuda"), status=torch.zeros(B, dtype=torch.int32, device="cuda"),
           iters=torch.zeros(B, dtype=torch.int32, device="cuda"))
step = bt.make_tick_call(dev, out, 0.002)

def timed():
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            step()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    return float(np.median(ts))

bt.set_option("presolve", 0)
t_gen = timed()
ref = out["qdot"].clone()
print(json.dumps({"path": "general kernel alone (presolve off)", "ms_per_step": t_gen, "M_ticks_per_s": B / t_gen / 1e3}), flush=True)
bt.set_option("presolve", 1)
# the packed kernel (four instances per wavefront): pivoted elimination + swap in place, nothing deferred
counts = {}
for tol in (7, 5, 3, 0):
    bt.set_option("presolve_tol_exp", tol)
    bt.set_option("packed_kernel", 0)
    bt.set_option("count_pivoted", 1)
    step()
    counts[tol] = bt.stat("pivoted_last")
    bt.set_option("count_pivoted", 0)
    bt.set_option("packed_kernel", 1)
    t = timed()
    assert bt.stat("last_path") == 2 and bt.stat("deferred_last") == 0
    err = float((out["qdot"] - ref).abs().max().item())
    print(json.dumps({"path": "packed kernel: pivoted elimination + swap in place", "presolve_tol_exp": tol, "flagged_instances": counts[tol],
                      "flagged_frac": counts[tol] / B, "ms_per_step": t, "M_ticks_per_s": B / t / 1e3, "vs_general_alone": t / t_gen,
                      "max_abs_diff_vs_general": err}), flush=True)
bt.set_option("packed_kernel", 0)
for defer in (0, 1):
    bt.set_option("dbg_force_defer", defer)
    for tol in (7, 5, 3, 0):
        bt.set_option("presolve_tol_exp", tol)
        bt.set_option("count_pivoted", 1)
        step()
        n = bt.stat("deferred_last") if defer else bt.stat("pivoted_last")
        bt.set_option("count_pivoted", 0)
        t = timed()
        err = float((out["qdot"] - ref).abs().max().item())
        print(json.dumps({"path": "one-instance compact kernel: second pass (general kernel over the compact list)" if defer else "one-instance compact kernel: pivoted elimination",
                          "presolve_tol_exp": tol, "flagged_instances": n, "flagged_frac": n / B, "ms_per_step": t, "M_ticks_per_s": B / t / 1e3,
                          "vs_general_alone": t / t_gen, "max_abs_diff_vs_general": err}), flush=True)
bt.close()
