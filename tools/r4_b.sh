#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4b; mkdir -p $O
python -m pytest tests/test_hip_knn.py -m gpu -x -q -k "matrix_pipe or g6 or larger" > $O/knn_tests.log 2>&1; echo "knn tests rc=$?"; tail -5 $O/knn_tests.log
python3 bench.py --workload knn --nobj 100000 --steps 3 --warmup 1 --no-cpu > $O/bench_knn.json 2> $O/bench_knn.err; echo "knn rc=$?"; cut -c1-300 $O/bench_knn.json
FZ_KNN_MORTON=1 python3 bench.py --workload knn --nobj 100000 --steps 3 --warmup 1 --no-cpu > $O/bench_knn_morton.json 2> $O/bench_knn_morton.err; echo "knn morton rc=$?"; cut -c1-300 $O/bench_knn_morton.json
for m in "" "--exact"; do
FZ_BENCH_NO_EXTRA=1 python3 bench.py --nobj 262144 --steps 3 --warmup 1 --no-cpu $m > $O/bench_hist$m.json 2> $O/bench_hist$m.err; echo "hist $m rc=$?"
python3 -c "
import json; d=json.loads(open('$O/bench_hist$m.json').read().strip().splitlines()[-1])
print('value %.4g %s  ms/step %.2f  frac %.3f form %s' % (d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d['config']['kernel_form']))"
done
python tools/parity_quick.py 2>&1 | tail -5
