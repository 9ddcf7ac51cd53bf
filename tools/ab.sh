#!/bin/bash
# same-box A/B of two library builds: FRANKENZ_HIP_LIB selects the library (tuning aid)
#   ./tools/ab.sh frankenz_amd/csrc/libfrankenz_hip_old.so
OLD=$1
run() { python3 bench.py --no-cpu --nobj 262144 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g' % d['value'], '%.1f' % d['ms_per_step'])"; }
for cfg in "--mode A" "--mode Ai" "--mode B" "--model-err varying" "--prior 64" "--mask-frac 0.02" "--noise-scale 3" "--noise-scale 10" "--mode B --noise-scale 3" "--nband 8" "--nband 12 --nobj 131072"; do
  a=$(FRANKENZ_HIP_LIB=$PWD/$OLD run $cfg); b=$(run $cfg); a2=$(FRANKENZ_HIP_LIB=$PWD/$OLD run $cfg); b2=$(run $cfg)
  echo "$cfg | old $a ; $a2 | new $b ; $b2"
done
