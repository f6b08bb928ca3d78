#!/bin/bash
# VGPR / SGPR / spill / LDS / scratch figures of every kernel in the built library: see tools/kernel_resources.py (one code object per translation unit).
exec python3 "$(dirname "$0")/kernel_resources.py" "$@"
