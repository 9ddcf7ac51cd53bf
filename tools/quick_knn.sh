#!/bin/bash
# k-NN search: kernel table of the shipped (matrix-pipe) build; pass "test" first to run the k-NN parity tests
export TMPDIR=/tmp
export FZ_BENCH_NO_EXTRA=1
O=gpurun_out
mkdir -p $O
if [ "$1" == "test" ]; then
  timeout -k 10 600 python -m pytest tests/test_hip_knn.py tests/test_hip_fullsize.py -m gpu -x -q > $O/knn_t1.log 2>&1 || { tail -30 $O/knn_t1.log; exit 1; }
  tail -2 $O/knn_t1.log
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $O/knn_stats -- python3 bench.py --workload knn --nobj 100000 --no-cpu --steps 3 --warmup 1 > $O/knn_stats.log 2>&1
tail -1 $O/knn_stats.log | cut -c1-200
f=$(ls -t $O/knn_stats/*/*kernel_stats.csv | head -1); head -3 $f | cut -c1-200
