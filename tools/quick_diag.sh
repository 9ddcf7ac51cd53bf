#!/bin/bash
# time variant libraries (built with tools/devbuild.sh -o <name>.so ...) on the headline shape, 262 144 objects
export TMPDIR=/tmp
export FZ_BENCH_NO_EXTRA=1
A="--nobj 262144 --no-cpu --steps 2 --warmup 1"
cp frankenz_amd/csrc/libfrankenz_hip.so /tmp/lib_keep.so
for lib in "$@"; do
  cp frankenz_amd/csrc/$lib frankenz_amd/csrc/libfrankenz_hip.so
  [ -n "$PARITY" ] && python3 tools/parity_quick.py 2>&1 | tail -1
  for extra in "" "--mode B" "--model-err varying"; do
  python3 bench.py $A $extra 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib', '$extra', '%.4g evals/s  %.2f ms' % (d['value'], d['ms_per_step']), d['pdfs_normalised'])"
  done
done
cp /tmp/lib_keep.so frankenz_amd/csrc/libfrankenz_hip.so
