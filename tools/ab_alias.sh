for i in 1 2 3; do
  for al in 0 1; do
    WBC_DBG_ALIAS=$al timeout -k 10 200 python bench.py --rollout-ticks 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.read()); print('alias $al', '%.2f M ticks/s' % (l['value']/1e6), ['%.4f' % x for x in l['repeats']['kernel_ms_per_step']], l['config']['kernel_path'][:30])"
  done
done
