#!/bin/bash
# per-kernel time of one bench configuration (rocprofv3 --kernel-trace --stats): tools/kstats.sh TAG [bench.py args]
#   LIB=libfz_dev.so selects a development library
export TMPDIR=/tmp
export FZ_BENCH_NO_EXTRA=1
[ -n "$LIB" ] && export FRANKENZ_HIP_LIB=$PWD/frankenz_amd/csrc/$LIB
TAG=$1; shift
O=gpurun_out/ks_$TAG; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --no-cpu "$@" > $O/bench.log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp $f $O/kernel_stats.csv; python3 - $O/kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print('%-90s calls %4s  avg %10.3f ms  total %10.3f ms  %5s %%' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e6, float(r['TotalDurationNs']) / 1e6, r['Percentage']))
PY
else echo "no kernel_stats.csv"; tail -5 $O/bench.log; fi
