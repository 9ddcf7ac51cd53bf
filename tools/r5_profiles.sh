#!/bin/bash
# the profile set behind a profiles/rN_vM_* row: rocprofv3 --kernel-trace --stats of the DEFAULT bench command (the driver's line) and the
# PMC passes (separate runs) of the headline / general / mode-B / catalogue launches at the bench's own shape
#   TAG=r5_v1 tools/r5_profiles.sh
export TMPDIR=/tmp
TAG=${TAG:-r5_v1}
O=gpurun_out/$TAG; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python3 bench.py --no-cpu --steps 3 --warmup 1 > $O/bench_default_line.json 2> $O/bench_default_line.err
f=$(find $O/stats_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_kernel_stats_default_line.csv
echo "stats done"
TAG=${TAG}_headline tools/pmc.sh > /dev/null 2>&1; echo "pmc headline done"
TAG=${TAG}_general BENCH_ARGS="--model-err varying" tools/pmc.sh > /dev/null 2>&1; echo "pmc general done"
TAG=${TAG}_modeB BENCH_ARGS="--mode B" tools/pmc.sh > /dev/null 2>&1; echo "pmc modeB done"
TAG=${TAG}_catalogue BENCH_ARGS="--model-err varying --mask-frac 0.02 --model-mask-frac 0.02" tools/pmc.sh > /dev/null 2>&1; echo "pmc catalogue done"
TAG=${TAG}_catalogue_widths BENCH_ARGS="--model-err varying --mask-frac 0.02 --model-mask-frac 0.02 --label-err varying" tools/pmc.sh > /dev/null 2>&1; echo "pmc catalogue widths done"
TAG=${TAG}_modeC NOBJ=20000 NMODEL=10000 BENCH_ARGS="--mode C --model-err varying" tools/pmc.sh > /dev/null 2>&1; echo "pmc mode C done"
ls gpurun_out/pmc_${TAG}_*.txt
