#!/bin/bash
# Resource usage of every gfx950 kernel in wbc_kernels.hip (VGPRs, spills, LDS, code bytes) — CPU only, no GPU needed.
set -e
cd "$(dirname "$0")/../mech5845m-wbc-for-legged-manipulator_amd/csrc"
mkdir -p build
hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function --cuda-device-only -S wbc_kernels.hip -o build/wbc_kernels.s "$@"
awk '/^_ZN3wbc.*:$/ {name=$1} /\.vgpr_count|\.vgpr_spill_count|\.sgpr_spill_count|\.group_segment_fixed_size|\.private_segment_fixed_size|\.name:/ {print}' build/wbc_kernels.s | paste - - - - - - | sed 's/ \+/ /g'
grep -E "; codeLenInByte" build/wbc_kernels.s
