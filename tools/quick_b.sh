#!/bin/bash
# mode B bodies A/B: weight-space (fp32 tail) vs ln-space (all fp64), 262144 objects x 1e5 models
run() { python3 bench.py --no-cpu --nobj 262144 --steps 2 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g evals/s  %.1f ms' % (d['value'], d['ms_per_step']))"; }
for cfg in "--mode B" "--mode B --nband 4" "--mode B --nband 6" "--mode Bn" "--mode An" ; do
  echo "$cfg | wspace $(run $cfg) | ln-space $(FZ_NO_WSPACE=1 run $cfg) | ln-space 4,8 $(FZ_NO_WSPACE=1 FZ_FUSED_CFG=4,8 run $cfg)"
done
