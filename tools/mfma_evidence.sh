#!/bin/bash
# Matrix-core evidence for the J'J contraction (QP_Wrapper.py:17-18), on the GPU box via gpurun:
# each case once under rocprofv3 --kernel-trace --pmc <MFMA counters> (program directly after --), MFMA on and off,
# throughput printed by the case itself. Output: gpurun_out/$1/{case logs, mfma_summary.txt}.
set -u
TAG=${1:-mfma}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {  # name, args...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv \
      -d "$OUT/$name" -- python3 tools/time_mfma.py "$@" > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "[$name] rc=$rc $(grep MFMA_CASE "$OUT/$name.log")"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "case $name was killed at its limit: stopping"; exit 1; fi
}
for mf in 0 1; do
  run tick_c2_$mf tick c2 65536 $mf
  run tick_c3_$mf tick c3 65536 $mf
  run tick_everything_$mf tick everything 65536 $mf
  for m in 32 64 96; do run qpls_m${m}_$mf qpls $m 65536 $mf; done
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
with open(os.path.join(out, "mfma_summary.txt"), "w") as w:
    for log in sorted(glob.glob(os.path.join(out, "*.log"))):
        name = os.path.basename(log)[:-4]
        case = [l.strip() for l in open(log) if l.startswith("MFMA_CASE")]
        agg = collections.defaultdict(float); n = collections.Counter()
        for f in glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row.get("Kernel_Name", "")
                if "wbc_tick_kernel" not in k and "wbc_qp_kernel" not in k: continue
                if "wbc_tick_kernel<2>" in k or "wbc_tick_kernel<1>" in k: continue      # the FK-only launches that place the inputs
                agg[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
        line = "%s | %s | " % (name, case[0] if case else "no timing line") + "  ".join(
            "%s/dispatch=%.4g" % (c, agg[c] / max(1, n[c])) for c in sorted(agg))
        print(line); w.write(line + "\n")
PY
