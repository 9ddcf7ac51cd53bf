#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
python -m pytest tests/test_hip_device_resident.py tests/test_hip_modec_shapes.py -m gpu -x -q > $O/new_tests.log 2>&1; echo "new tests rc=$?"; tail -15 $O/new_tests.log
python -m pytest tests -m gpu -q --deselect tests/test_hip_device_resident.py --deselect tests/test_hip_modec_shapes.py > $O/gputest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/gputest.log
python3 bench.py --steps 3 --warmup 1 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python3 -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('value %.4g %s  ms/step %.2f  frac %.3f form %s' % (d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d['config']['kernel_form']))
for k in ('roofline_fp64','roofline_general','roofline_modeB'): print(k, '%.4g' % d[k]['value'], '%.3f' % d[k]['frac'], d[k]['kernel'])"
python3 bench.py --workload knn --nobj 100000 --steps 3 --warmup 1 --no-cpu > $O/bench_knn.json 2> $O/bench_knn.err; echo "knn rc=$?"; cut -c1-300 $O/bench_knn.json
FZ_KNN_MORTON=1 python3 bench.py --workload knn --nobj 100000 --steps 3 --warmup 1 --no-cpu > $O/bench_knn_morton.json 2> $O/bench_knn_morton.err; echo "knn morton rc=$?"; cut -c1-300 $O/bench_knn_morton.json
python tools/host_path_timing.py > $O/host_path.txt 2>&1; cat $O/host_path.txt
