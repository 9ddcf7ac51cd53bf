#!/bin/bash
# geometry A/B: FZ_FUSED_CFG per mode (262144 objects x 1e5 models)
run() { python3 bench.py --no-cpu --nobj 262144 --steps 2 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g evals/s  %.1f ms' % (d['value'], d['ms_per_step']))"; }
for cfg in "--model-err varying" "--mode B" "--nband 6" "--nband 7" "--nband 8" "--nband 4" "--mode B --nband 4"; do
  for g in 2,16 2,12 4,8; do
    echo "$cfg | cfg $g | $(FZ_FUSED_CFG=$g run $cfg)"
  done
done
