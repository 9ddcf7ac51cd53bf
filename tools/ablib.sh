#!/bin/bash
# A/B of library builds in one gpurun call: tools/devbuild.sh -o libfz_x.so -DSOMETHING (here), then
#   gpurun -- 'tools/ablib.sh "--mode A" libfrankenz_hip.so libfz_x.so'
#   gpurun -- 'NOBJ=100000 tools/ablib.sh "--workload knn" libfz_a.so libfz_b.so'
export TMPDIR=/tmp; export FZ_BENCH_NO_EXTRA=1
ARGS=$1; shift
for rep in 1 2; do for lib in "$@"; do
  FRANKENZ_HIP_LIB=$PWD/frankenz_amd/csrc/$lib python3 bench.py --nobj ${NOBJ:-262144} --steps ${STEPS:-4} --warmup 1 --no-cpu $ARGS 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-28s %.4g %s  %.2f ms/step' % ('$lib', d['value'], d['unit'], d['ms_per_step']))"
done; done
