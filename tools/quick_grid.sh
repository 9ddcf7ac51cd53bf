#!/bin/bash
# direct gauss_kde (grid labels): parity of the fused path against the oracle, then the bench line with and without the recurrence
export FZ_BENCH_NO_EXTRA=1
PQ_KDE=grid python3 tools/parity_quick.py 2>&1 | tail -6
python3 -m pytest tests -m gpu -x -q -k "grid or gauss or kde or window" --tb=short 2>&1 | tail -3
for env in "FZ_GRID_RECUR=1" "FZ_GRID_RECUR=0"; do
  env $env python3 bench.py --kde grid --nobj 262144 --steps 2 --warmup 1 --no-cpu 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$env grid KDE: %.3e evals/s, %.1f ms/step, form %s' % (d['value'], d['ms_per_step'], d['config'].get('kernel_form')))"
done
