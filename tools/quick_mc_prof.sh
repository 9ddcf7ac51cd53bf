#!/bin/bash
export TMPDIR=/tmp
export FZ_BENCH_NO_EXTRA=1
O=gpurun_out
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/mc_stats -- python3 bench.py --label-err varying --nobj 262144 --no-cpu --steps 2 --warmup 1 > $O/mc_stats.log 2>&1
f=$(ls -t $O/mc_stats/*/*kernel_stats.csv | head -1); head -6 $f | cut -c1-250
