#!/bin/bash
# wide-band fit_predict lines (12 / 16 / 32 bands), 262 144 objects x 1e5 models
export FZ_BENCH_NO_EXTRA=1
for nb in ${BANDS:-12 16 32}; do
  python3 bench.py --nband $nb --nobj ${NOBJ:-262144} --steps 2 --warmup 1 --no-cpu $EXTRA 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('nband %s: %.3e evals/s, %.1f ms/step, form %s' % ('$nb', d['value'] if 'evals' in d['unit'] else d.get('evals_per_s', 0), d['ms_per_step'], d['config'].get('kernel_form')))"
done
