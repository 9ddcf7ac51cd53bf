#!/bin/bash
# objects with unobserved bands: parity subset (TESTS=1), then the bench line with per-object band counts on k_hist and with the
# split into mask-free / masked launches (FZ_HIST_OBJMASK=0), at two masked fractions, modes A (constant errors) and B
export FZ_BENCH_NO_EXTRA=1
if [ -n "$TESTS" ]; then python3 -m pytest tests -m gpu -x -q -k "${TESTK:-mask or golden or g1_ or g3 or g4}" --tb=short 2>&1 | tail -6; fi
for frac in 0.02 0.2; do
  for extra in "" "--mode B"; do
    for e in "FZ_HIST_OBJMASK=1" "FZ_HIST_OBJMASK=0"; do
      env $e python3 bench.py --mask-frac $frac --nobj ${NOBJ:-262144} --steps 2 --warmup 1 --no-cpu $extra 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('mask-frac $frac $extra $e: %.3e evals/s, %.1f ms/step, form %s' % (d['value'], d['ms_per_step'], d['config'].get('kernel_form')))"
    done
  done
done
