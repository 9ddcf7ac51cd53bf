#!/bin/bash
# rocprofv3 kernel statistics of the secondary workloads (writes gpurun_out/x_*)
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/x_stats_modeB -- python3 bench.py --mode B --no-cpu --steps 2 --warmup 1 > $O/x_stats_modeB.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/x_stats_planes -- python3 bench.py --workload fit --nobj 100000 --nmodel 10000 --no-cpu --steps 10 --warmup 2 > $O/x_stats_planes.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/x_stats_knn -- python3 bench.py --workload knn --nobj 100000 --no-cpu --steps 2 --warmup 1 > $O/x_stats_knn.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/x_stats_varying -- python3 bench.py --model-err varying --no-cpu --steps 2 --warmup 1 > $O/x_stats_varying.log 2>&1
