#!/bin/bash
# FETCH_SIZE and WRITE_SIZE (separate rocprofv3 passes: the two do not fit one pass on gfx950) of tools/calib_traffic's known-byte-count
# kernels, then of the bench command's tick kernel; prints counter / true-bytes per kernel and the calibrated traffic of the tick kernel.
# usage (on the GPU box): bash tools/calib_traffic.sh [tag]
set -u
TAG=${1:-calib}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
step() { local name=$1 t=$2; shift 2; timeout -k 10 "$t" "$@" > "$OUT/$name.log" 2>&1; local rc=$?; echo "[$name] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed at its limit: stopping"; exit 1; fi; }
BENCH="python3 bench.py --steps 4 --warmup 1 --repeats 1 --rollout-ticks 0 --no-cpu-baseline"
step calib_fetch 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/calib_fetch" -- tools/build/calib_traffic 65536 5
step calib_write 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/calib_write" -- tools/build/calib_traffic 65536 5
step tick_fetch 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/tick_fetch" -- $BENCH
step tick_write 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/tick_write" -- $BENCH
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
true = json.loads([l for l in open(os.path.join(out, "calib_fetch.log")) if l.startswith("{")][-1])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        name = next((n for n in ("calib_stream16", "calib_stream8", "calib_tick_rows", "wbc_tick_sim3p_kernel") if n in k), None)
        if name:
            agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
lines = []
fac = {}
for name in ("calib_stream16", "calib_stream8", "calib_tick_rows"):
    fs = sum(agg[name]["FETCH_SIZE"]) / max(1, len(agg[name]["FETCH_SIZE"])) * 1024.0
    ws = sum(agg[name]["WRITE_SIZE"]) / max(1, len(agg[name]["WRITE_SIZE"])) * 1024.0
    t = true[name]
    fac[name] = (t["read_bytes"] / fs if fs else float("nan"), t["write_bytes"] / ws if ws else float("nan"))
    lines.append("%-16s true read %10d B  FETCH_SIZE %12.0f B  (true / counter = %.3f)   true write %10d B  WRITE_SIZE %12.0f B  (true / counter = %.3f)" % (
        name, t["read_bytes"], fs, fac[name][0], t["write_bytes"], ws, fac[name][1]))
k = "wbc_tick_sim3p_kernel"
if agg[k]["FETCH_SIZE"]:
    fs = sum(agg[k]["FETCH_SIZE"]) / len(agg[k]["FETCH_SIZE"]) * 1024.0
    ws = sum(agg[k]["WRITE_SIZE"]) / len(agg[k]["WRITE_SIZE"]) * 1024.0
    fr, fw = fac["calib_tick_rows"]
    lines.append("%-16s FETCH_SIZE %12.0f B  WRITE_SIZE %12.0f B per dispatch (B = 65536);  calibrated with calib_tick_rows' factors: read %.0f B + write %.0f B = %.0f B "
                 "= %.1f B per tick (streamed by the kernel: 704 B per tick, algorithmic: 572 B)" % (k, fs, ws, fs * fr, ws * fw, fs * fr + ws * fw, (fs * fr + ws * fw) / 65536.0))
    lines.append("CALIBRATED_TRAFFIC_BYTES_PER_DISPATCH %.0f" % (fs * fr + ws * fw))
open(os.path.join(out, "calib_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
