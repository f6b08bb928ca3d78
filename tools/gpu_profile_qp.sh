#!/bin/bash
# rocprofv3 kernel-trace stats + PMC passes (one --pmc group per run) of the stand-alone QP at (m, n, p) = (32, 26, 16): python3 tools/time_qp.py ls.
# Output under gpurun_out/$1; a killed step stops the script.   usage: tools/gpu_profile_qp.sh [tag]
set -u
TAG=${1:-prof_qp}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp

CMD="python3 tools/time_qp.py ls"
step() { local name=$1 t=$2; shift 2; timeout -k 10 "$t" "$@" > "$OUT/$name.log" 2>&1; local rc=$?; echo "[$name] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed at its limit: stopping"; exit 1; fi; }
step stats 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD
pmc() { step "pmc_$1" 200 rocprofv3 --kernel-trace --pmc ${@:2} --output-format csv -d "$OUT/pmc_$1" -- $CMD; }
pmc inst SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM
pmc cyc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pmc mfma SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950: 3 + 2 of the 4 TCC slots)
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); grid = {}
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "wbc_qp" not in k: continue
        g = row.get("Grid_Size", "")
        key = (k, g)
        agg[key][row["Counter_Name"]] += float(row["Counter_Value"]); calls[(key, row["Counter_Name"])] += 1
with open(os.path.join(out, "pmc_summary_qp.txt"), "w") as w:
    for key, d in sorted(agg.items()):
        for c, v in sorted(d.items()):
            line = "%s grid=%s %s total=%.6g dispatches=%d per_dispatch=%.6g" % (key[0][:60], key[1], c, v, calls[(key, c)], v / calls[(key, c)])
            print(line); w.write(line + "\n")
PY
