#!/usr/bin/env python3
"""Closed-loop roll-outs (wbc_rollout) with and without the carried working set, on the stressed (C3 recipe) and
the un-stressed input distribution: ms per closed-loop tick, working-set changes per tick, worst-status histogram.
    python3 tools/time_rollout.py [B] [K]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mech5845m-wbc-for-legged-manipulator_amd"))
import numpy as np, torch
import wbc_model, wbc_workload
from wbc_batch import WbcBatch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
model = wbc_model.load_model("a1_wx200")
cfg = wbc_model.sim3_config(model)
bt = WbcBatch(model, B)
bt.configure(cfg)
fk = lambda q: bt.fk(q, want=("oMf",))["oMf"]
out = []
for stress in (True, False):
    d = wbc_workload.make_tick_inputs(model, cfg, B, 0, fk, stress=stress)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items()}
    step = torch.zeros((B, 5, 3), dtype=torch.float64, device="cuda")
    step[:, 4, 0] = 1e-4
    for warm in (0, 1):
        bt.set_option("warm_start", warm)
        bt.rollout(dev, 0.002, 2, ee_target_step=step, want_trace=False)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ro = bt.rollout(dev, 0.002, K, ee_target_step=step, want_trace=False)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / K)
        st = ro["status"].cpu().numpy()
        rec = {"inputs": "stressed" if stress else "unstressed", "warm_start": warm, "ms_per_tick": float(np.median(ts)),
               "M_closed_loop_ticks_per_s": B / float(np.median(ts)) / 1e3, "working_set_changes_per_tick": float(ro["iters"].double().mean().item()) / K,
               "worst_status": np.bincount(st, minlength=4).tolist()}
        out.append(rec)
        print(json.dumps(rec), flush=True)
bt.close()

# ---- (c) open-loop ticks on the same inputs: cold vs hot-started with the previous tick's / the tick's own final working set (the steady
# state of a smooth trajectory), kernel time by HIP events; "slowest row per wave" = max over the four instances of a wavefront of the
# inequality working-set changes (iters - eliminated equalities - locked DoF): the passes the packed kernel's loops run
bt2 = WbcBatch(model, B)
bt2.configure(cfg)
fk2 = lambda q: bt2.fk(q, want=("oMf",))["oMf"]
for stress in (True, False):
    d = wbc_workload.make_tick_inputs(model, cfg, B, 0, fk2, stress=stress)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items()}
    prev = dict(dev)
    prev["ee_target"] = dev["ee_target"] - 1e-4
    ws_prev = bt2.tick(prev, 0.002, want_working_set=True)["working_set"]
    cold = bt2.tick(dev, 0.002, want_working_set=True)
    base = int(cold["iters"].min().item())
    for name, ws in (("cold", None), ("own set", cold["working_set"]), ("previous tick's set", ws_prev)):
        o = dict(qdot=torch.zeros((B, 26), dtype=torch.float64, device="cuda"), status=torch.zeros(B, dtype=torch.int32, device="cuda"),
                 iters=torch.zeros(B, dtype=torch.int32, device="cuda"))
        inp = dict(dev) if ws is None else dict(dev, working_set=ws)
        if ws is not None:
            o["working_set"] = torch.zeros((B, 2), dtype=torch.int64, device="cuda")
        call = bt2.make_tick_call(inp, o, 0.002)
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(K):
                call()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / K)
        it = (o["iters"].cpu().numpy().astype(np.int64) - base).clip(min=0)
        ok = o["status"].cpu().numpy() == 0
        per_wave = it[: B // 4 * 4].reshape(-1, 4).max(axis=1)
        rec = {"inputs": "stressed" if stress else "unstressed", "open_loop_tick": name, "kernel_path": int(bt2.stat("last_path")), "ms_per_tick": float(np.median(ts)),
               "M_ticks_per_s": B / float(np.median(ts)) / 1e3, "ineq_changes_per_instance": float(it[ok].mean()),
               "slowest_row_per_wave_mean": float(per_wave.mean()), "slowest_row_per_wave_p99": float(np.percentile(per_wave, 99)),
               "err_vs_cold": float((o["qdot"] - cold["qdot"]).abs()[torch.from_numpy(ok).cuda()].max().item())}
        out.append(rec)
        print(json.dumps(rec), flush=True)
bt2.close()
