#!/bin/bash
# mode C: parity (golden G2 + the mode C parametrisations of the suite), then the bench line with the reciprocal-based and the IEEE solve
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "mode_c or modec or g2 or cdf_threshold or g1_" 2>&1 | tail -3
for cfg in "" "FZ_MODEC_IEEE=1" "FZ_MODEC_PLANES=1"; do
  env $cfg python3 bench.py --mode C --model-err varying --nobj 20000 --nmodel 10000 --no-cpu --steps 2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg', '%.4g evals/s  %.2f ms/step  iterations %s' % (d['value'], d['ms_per_step'], d['roofline']['modec_iterations_per_step']))"
done
