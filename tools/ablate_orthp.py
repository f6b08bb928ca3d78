#!/usr/bin/env python3
"""Where the packed orth kernel's time goes (BASELINE configs[1] at B = 65536; or, second argument, tests/common.py "everything": its INEQ variant), by ablation: option "dbg_stop" = 200 + k of the -DWBC_ABLATE
build cuts wbc_tick_orthp_kernel after stage k; stage k costs T(k) - T(k - 1). Same occupancy as the product; read shares.
    python3 tools/ablate_orthp.py [B] [c2 | everything]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mech5845m-wbc-for-legged-manipulator_amd")
sys.path.insert(0, PKG)
subprocess.check_call(["make", "-s", "-C", os.path.join(PKG, "csrc"), "ablate"])
os.environ["WBC_HIP_LIB"] = os.path.join(PKG, "csrc", "build", "libwbc_hip_ablate.so")
import numpy as np, torch
import wbc_model, wbc_workload
from wbc_batch import WbcBatch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
name = sys.argv[2] if len(sys.argv) > 2 else "c2"          # "everything": the kernel's INEQ variant (the cuts are the same stages)
model = wbc_model.load_model("a1_wx200")
if name == "c2":
    cfg = wbc_model.equality_only_config(model)
else:
    sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
    import common
    cfg = common.config(name, model)
bt = WbcBatch(model, B)
bt.configure(cfg)
bt.set_option("packed_orth", 2)      # (also below the batch-size policy's threshold: B = 1024 shows one wave's latency, stage by stage)

class FK:
    def __call__(_, q):
        return bt.fk(q, want=("oMf",))["oMf"]
    def com(_, q):
        return bt.fk(q, want=("com",))["com"]
d = wbc_workload.make_tick_inputs(model, cfg, B, 5, FK()) if name == "c2" else common.tick_inputs(model, cfg, B, 5, with_rot=True)
dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items()}
out = dict(qdot=torch.zeros((B, 26), dtype=torch.float64, device="cuda"), status=torch.zeros(B, dtype=torch.int32, device="cuda"),
           iters=torch.zeros(B, dtype=torch.int32, device="cuda"))
step = bt.make_tick_call(dev, out, 0.002)
names = {201: "inputs, weights and plan records staged, sin / cos, root placement", 202: "FK levels (6)", 203: "frames, m c, Jacobian columns, CoM + its Jacobian columns",
         204: "contact rows, G, M = I + G'G, cooperative Cholesky, Z", 205: "task blocks (CoM + 5 EE): images, A Z, H' and g'", 206: "posture + Cholesky sweep with substitutions",
         0: "y, qdot = Z y, outputs (+ tail for flagged instances)"}
res, prev = {}, 0.0
for k in (201, 202, 203, 204, 205, 206, 0):
    bt.set_option("dbg_stop", k)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    assert bt.stat("last_path") == 3
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            step()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    t = float(np.median(ts))
    res[names[k]] = {"stop": k, "cumulative_ms": t, "stage_ms": t - prev}
    prev = t
    print("stop %3d  %-75s cumulative %.4f ms  stage %.4f ms" % (k, names[k], t, res[names[k]]["stage_ms"]), flush=True)
bt.set_option("dbg_stop", 0)
print(json.dumps({"B": B, "stages": res}))
bt.close()
