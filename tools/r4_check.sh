#!/bin/bash
# round-4 regression of what changed outside the kernels: new GPU tests first, then (FULL=1) the whole suite, the PCIe-inclusive
# drop-in timing, the k-NN line through the drop-in class, and 2 gloo ranks on the one GPU through bench.py (k-NN sharded path)
export TMPDIR=/tmp
O=gpurun_out/r4check; mkdir -p $O
python -m pytest tests/test_hip_device_resident.py tests/test_hip_modec_shapes.py tests/test_hip_sharded.py -m gpu -x -q > $O/new_tests.log 2>&1; echo "new tests rc=$?"; tail -5 $O/new_tests.log
if [ -n "$FULL" ]; then python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/gputest.log; fi
python tools/host_path_timing.py > $O/host_path.txt 2>&1; cat $O/host_path.txt
python3 bench.py --workload knn --nobj 100000 --steps 3 --warmup 1 --no-cpu > $O/bench_knn.json 2> $O/bench_knn.err; echo "knn rc=$?"; cut -c1-400 $O/bench_knn.json
FZ_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --workload knn --nobj 100000 --steps 2 --warmup 1 --no-cpu > $O/bench_knn_2ranks_gloo.json 2> $O/bench_knn_2ranks.err; echo "knn 2 ranks rc=$?"; cut -c1-600 $O/bench_knn_2ranks_gloo.json
FZ_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --nobj 200000 --steps 2 --warmup 1 --no-cpu > $O/bench_2ranks_gloo.json 2> $O/bench_2ranks.err; echo "fit_predict 2 ranks rc=$?"; cut -c1-300 $O/bench_2ranks_gloo.json
