#!/bin/bash
# Same-box A/B of the shipped library against variant builds (make -C csrc variant VFLAGS=... [VSUF=2]): bench.py three times each, interleaved.
# Prints value and kernel ms per step. AB_LIBS="variant variant2" picks the variants (default: variant).
D=mech5845m-wbc-for-legged-manipulator_amd/csrc/build
for i in 1 2 3; do
  for v in shipped ${AB_LIBS:-variant}; do
    lib=""; [ "$v" != shipped ] && lib=$D/libwbc_hip_$v.so
    WBC_HIP_LIB=$lib timeout -k 10 200 python bench.py --rollout-ticks 0 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.read()); print('$v', '%.2f M ticks/s' % (l['value']/1e6), ['%.4f' % x for x in l['repeats']['kernel_ms_per_step']], 'err %s' % l.get('accuracy',{}).get('qdot_max_abs_err_vs_cpu'))"
  done
done
