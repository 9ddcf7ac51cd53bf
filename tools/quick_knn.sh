#!/bin/bash
# k-NN search: kernel table of the shipped (matrix-pipe) build; pass "test" first to run the k-NN parity tests;
# remaining arguments: FZ_KNN_GEOM values to time
export TMPDIR=/tmp
export FZ_BENCH_NO_EXTRA=1
O=gpurun_out
mkdir -p $O
if [ "$1" == "test" ]; then
  shift
  for g in "$@"; do
  FZ_KNN_GEOM=$g timeout -k 10 600 python -m pytest tests/test_hip_knn.py tests/test_hip_fullsize.py -m gpu -x -q -k "knn" > $O/knn_t$g.log 2>&1 || { tail -30 $O/knn_t$g.log; exit 1; }
  tail -1 $O/knn_t$g.log
  done
fi
for g in "$@"; do
export FZ_KNN_GEOM=$g
rocprofv3 --kernel-trace --stats --output-format csv -d $O/knn_stats$g -- python3 bench.py --workload knn --nobj 100000 --no-cpu --steps 3 --warmup 1 > $O/knn_stats$g.log 2>&1
tail -1 $O/knn_stats$g.log | cut -c1-200
f=$(ls -t $O/knn_stats$g/*/*kernel_stats.csv | head -1); head -3 $f | cut -c1-200
done
