#!/bin/bash
# k-NN: parity tests, then the bench line with row-parallel and with wave-serial admissions
export TMPDIR=/tmp
O=gpurun_out/knn; mkdir -p $O
python -m pytest tests/test_hip_knn.py tests/test_hip_fullsize.py -m gpu -x -q -k "knn" 2>&1 | tail -4
for cfg in "" "FZ_KNN_SERIAL=1"; do
  env $cfg python3 bench.py --workload knn --nobj 100000 --no-cpu --steps 3 --warmup 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg', '%.4g objects/s  %.2f ms/step  search kernels %.2f ms' % (d['value'], d['ms_per_step'], d['kernel_ms_per_step']))"
done
