#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes of the bench command, each pass its own
# rocprofv3 run (counters never combined with trace domains other than kernel-trace). Output under gpurun_out/$1.
# A step that hits its timeout aborts the script (no further GPU work after a kill).
set -u
TAG=${1:-prof}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 4 --warmup 1 --repeats 1 --rollout-ticks 0 --no-cpu-baseline ${BENCH_ARGS:-}"
step() {  # name, timeout, command...
  local name=$1 t=$2; shift 2
  timeout -k 10 "$t" "$@" > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "[$name] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed at its limit: stopping"; exit 1; fi
}
step list 60 rocprofv3 -L
# kernel time: the bench command in its default shape (3 warm-up + 5 x 20 timed steps), so that the kernel average covers the same launches
# as bench.py's own HIP events (stats.log holds that run's JSON line: roofline.kernel_ms = the median block, repeats.kernel_ms_per_step = all)
step stats 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --rollout-ticks 0 --no-cpu-baseline ${BENCH_ARGS:-}
pmc() { step "pmc_$1" 200 rocprofv3 --kernel-trace --pmc ${@:2} --output-format csv -d "$OUT/pmc_$1" -- $BENCH; }
pmc inst SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM
pmc cyc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES
# (FETCH_SIZE / WRITE_SIZE: tools/calib_traffic.sh — calibrated against known-byte-count kernels in the tick's own access pattern)
pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
# condense: per-counter sums for the tick kernel
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "wbc_tick" not in k: continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[(k, row["Counter_Name"])] += 1
with open(os.path.join(out, "pmc_summary.txt"), "w") as w:
    for k, d in agg.items():
        for c, v in sorted(d.items()):
            line = "%s %s total=%.6g dispatches=%d per_dispatch=%.6g" % (k, c, v, calls[(k, c)], v / calls[(k, c)])
            print(line); w.write(line + "\n")
PY
