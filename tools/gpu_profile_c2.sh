#!/bin/bash
# rocprofv3 kernel-trace stats + PMC passes (one --pmc group per run) of one configuration of tools/time_configs.py (default c2 =
# BASELINE configs[1]: B = 1024 / 4096 on the general kernel's ORTH variant, B = 65536 on the packed orth kernel). Output under
# gpurun_out/$1; a killed step stops the script.   usage: tools/gpu_profile_c2.sh [tag] [configuration]
set -u
TAG=${1:-prof_c2}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CFG=${2:-c2}
CMD="python3 tools/time_configs.py $CFG"
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
        if "wbc_tick" not in k: continue
        g = row.get("Grid_Size", "")
        key = (k, g)
        agg[key][row["Counter_Name"]] += float(row["Counter_Value"]); calls[(key, row["Counter_Name"])] += 1
with open(os.path.join(out, "pmc_summary_c2.txt"), "w") as w:
    for key, d in sorted(agg.items()):
        for c, v in sorted(d.items()):
            line = "%s grid=%s %s total=%.6g dispatches=%d per_dispatch=%.6g" % (key[0][:60], key[1], c, v, calls[(key, c)], v / calls[(key, c)])
            print(line); w.write(line + "\n")
PY
