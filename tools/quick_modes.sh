#!/bin/bash
# one bench line per kernel family (262144 objects x 1e5 models), value / ms per step
run() { python3 bench.py --no-cpu --nobj 262144 --steps 2 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g evals/s  %.1f ms' % (d['value'], d['ms_per_step']))"; }
for cfg in "--mode A" "--mode Ai" "--mode B" "--model-err varying" "--prior 64" "--mask-frac 0.02" "--noise-scale 3" "--noise-scale 10" "--nband 8" "--label-err varying" "--kde grid"; do
  echo "$cfg | $(run $cfg)"
done
