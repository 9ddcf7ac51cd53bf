#!/bin/bash
export FZ_BENCH_NO_EXTRA=1
python3 -m pytest tests -m gpu -x -q -k "grid or gauss or kde" --tb=short 2>&1 | tail -5
for env in "FZ_GRID_RECUR=1" "FZ_GRID_RECUR=0"; do
  env $env python3 bench.py --kde grid --nobj 262144 --steps 2 --warmup 1 --no-cpu 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$env grid KDE: %.3e evals/s, %.1f ms/step, form %s' % (d['value'], d['ms_per_step'], d['config'].get('kernel_form')))"
done
