#!/bin/bash
# Same-box A/B of the refinement's cost on the benchmark: bench.py with WBC_REFINE=0 / 1, three times each, interleaved.
for i in 1 2 3; do
  for rf in 0 1; do
    WBC_REFINE=$rf timeout -k 10 200 python bench.py --rollout-ticks 0 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.read()); print('refine $rf', '%.2f M ticks/s' % (l['value']/1e6), ['%.4f' % x for x in l['repeats']['kernel_ms_per_step']])"
  done
done
