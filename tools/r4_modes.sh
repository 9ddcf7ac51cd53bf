#!/bin/bash
# one line per (likelihood mode, evidence form) at 262 144 objects: classifier form vs every-pair-in-fp64 form
export TMPDIR=/tmp
export FZ_BENCH_NO_EXTRA=1
O=gpurun_out/r4modes; mkdir -p $O
for cfg in "A_const:--mode A" "A_varying:--mode A --model-err varying" "B:--mode B" "Ai:--mode Ai"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  for ex in "" "--exact"; do
    python3 bench.py --nobj 262144 --steps 3 --warmup 1 --no-cpu $args $ex > $O/$tag$ex.json 2> $O/$tag$ex.err
    python3 -c "
import json; d=json.loads(open('$O/$tag$ex.json').read().strip().splitlines()[-1])
print('%-10s %-8s value %.4g  ms/step %.2f  frac %.3f form %s' % ('$tag', '$ex', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['kernel_form']))"
  done
done
