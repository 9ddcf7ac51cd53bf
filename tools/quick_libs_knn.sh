#!/bin/bash
export TMPDIR=/tmp
cp frankenz_amd/csrc/libfrankenz_hip.so /tmp/lib_keep.so
for lib in "$@"; do
  cp frankenz_amd/csrc/$lib frankenz_amd/csrc/libfrankenz_hip.so
  python3 bench.py --workload knn --nobj 100000 --no-cpu --steps 3 --warmup 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib', '%.4g objects/s  %.2f ms/step  search kernels %.2f ms' % (d['value'], d['ms_per_step'], d['kernel_ms_per_step']))"
done
cp /tmp/lib_keep.so frankenz_amd/csrc/libfrankenz_hip.so
