#!/bin/bash
# VGPR / SGPR / spill / LDS / scratch figures of every kernel in the built library (read from the code object's metadata notes).
# usage: tools/kernel_resources.sh [path/to/libwbc_hip.so]
set -e
SO=${1:-$(dirname "$0")/../mech5845m-wbc-for-legged-manipulator_amd/csrc/build/libwbc_hip.so}
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
objcopy -O binary --only-section=.hip_fatbin "$SO" "$TMP/fat.bin"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input="$TMP/fat.bin" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$TMP/co"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$TMP/co"* | python3 -c '
import sys, re
txt = sys.stdin.read()
for blk in re.split(r"\n\s*- \.agpr_count", txt)[1:]:
    def f(k):
        m = re.search(r"\." + k + r":\s*(\S+)", blk)
        return m.group(1) if m else "?"
    name = f("name")
    print("%-75s vgpr %3s (spill %3s)  sgpr %3s (spill %3s)  lds %6s  scratch %5s" % (name[:75], f("vgpr_count"), f("vgpr_spill_count"), f("sgpr_count"), f("sgpr_spill_count"), f("group_segment_fixed_size"), f("private_segment_fixed_size")))
'
