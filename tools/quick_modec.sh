#!/bin/bash
export TMPDIR=/tmp
export FZ_BENCH_NO_EXTRA=1
O=gpurun_out
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/modec_stats -- python3 bench.py --mode C --model-err varying --nobj 20000 --nmodel 10000 --no-cpu --steps 2 --warmup 1 > $O/modec_stats.log 2>&1
tail -1 $O/modec_stats.log | cut -c1-300
f=$(ls -t $O/modec_stats/*/*kernel_stats.csv | head -1); head -8 $f | cut -c1-200
