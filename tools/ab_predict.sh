#!/bin/bash
# A/B of library builds on the predict-from-a-stored-plane workload (three row lengths), two rounds, one gpurun call:
#   gpurun -- 'tools/ab_predict.sh libfrankenz_hip.so libfz_x.so'
export TMPDIR=/tmp
SHAPES=("100000 10000" "200000 5000" "50000 20000")
for rep in 1 2; do for sh in "${SHAPES[@]}"; do for lib in "$@"; do
  read NO NM <<< "$sh"
  FRANKENZ_HIP_LIB=$PWD/frankenz_amd/csrc/$lib python3 bench.py --workload predict --nobj $NO --nmodel $NM --steps 5 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-24s %6d x %6d  %.3f ms  %.0f GB/s  %s' % ('$lib', $NO, $NM, d['kernel_ms_per_step']['fused'], d['roofline']['achieved'], d['roofline']['kernel']))"
done; done; done
