#!/bin/bash
# no dimensionality prior (modes Ai / B: the power-0 form of k_hist): parity, then the bench line with and without it
export FZ_BENCH_NO_EXTRA=1
python3 -m pytest tests -m gpu -q --tb=line -k "not fullsize" 2>&1 | tail -15
for mode in An Bn; do
  for e in "FZ_HIST_NODIMPRIOR=1" "FZ_HIST_NODIMPRIOR=0"; do
    env $e python3 bench.py --mode $mode --nobj 262144 --steps 2 --warmup 1 --no-cpu 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('mode $mode $e: %.3e evals/s, %.1f ms/step, form %s' % (d['value'], d['ms_per_step'], d['config'].get('kernel_form')))"
  done
done
