#!/bin/bash
export TMPDIR=/tmp FZ_BENCH_NO_EXTRA=1
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "tuning_switches or exact_evidence" 2>&1 | tail -2
for extra in "--nmodel 1000000 --nobj 100000" "--nmodel 1000000 --nobj 100000 --mode B" "--nmodel 300000 --nobj 262144"; do
python3 bench.py $extra --no-cpu --steps 2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$extra', '%.4g evals/s  %.2f ms' % (d['value'], d['ms_per_step']), d['config']['kernel_form'], d['pdfs_normalised'])"
done
