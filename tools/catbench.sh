#!/bin/bash
# the catalogue configurations of the one-pass kernel (segmented form), one line each; LIB=libfz_dev.so tools/catbench.sh [NOBJ]
export TMPDIR=/tmp
[ -n "$LIB" ] && export FRANKENZ_HIP_LIB=$PWD/frankenz_amd/csrc/$LIB
N=${1:-262144}
for a in "--model-err varying --mask-frac 0.02 --model-mask-frac 0.02" "--model-err varying --mask-frac 0.02" "--model-err varying" "--mask-frac 0.02 --model-mask-frac 0.02" "--mode B --mask-frac 0.02 --model-mask-frac 0.02" "--mask-frac 0.02" ""; do
python3 bench.py --no-cpu --nobj $N --steps 3 $a 2>&1 | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-70s %.4g  %.1f ms  frac %.3f  %s norm=%s' % ('$a', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['kernel_form'], d['pdfs_normalised']))"
done
