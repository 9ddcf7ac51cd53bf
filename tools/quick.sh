#!/bin/bash
# One bench line per configuration, all in ONE gpurun call (boxes differ by ~8 %: only numbers of one call compare).
#   tools/quick.sh [-n NOBJ] [-s STEPS] [-t "pytest -k expr"] CONFIG...
# CONFIG = "[ENV=val ...] [bench.py args]", e.g.
#   tools/quick.sh "" "--mode B" "--mode Ai" "--model-err varying" "FZ_EXACT_EVIDENCE=1"       # likelihood modes / forms
#   tools/quick.sh -n 100000 "--workload knn" "FZ_KNN_SERIAL=1 --workload knn"                  # k-NN variants
#   tools/quick.sh "--nband 12" "--nband 24" "--mask-frac 0.2" "--kde grid" "--label-err varying" "--noise-scale 3"
#   tools/quick.sh -n 20000 "--nmodel 10000 --mode C"
export TMPDIR=/tmp
export FZ_BENCH_NO_EXTRA=1
NOBJ=262144; STEPS=3
while [[ $1 == -n || $1 == -s || $1 == -t ]]; do
  case $1 in -n) NOBJ=$2;; -s) STEPS=$2;; -t) TESTS=$2;; esac; shift 2
done
mkdir -p gpurun_out
if [ -n "$TESTS" ]; then python -m pytest tests -m gpu -x -q -k "$TESTS" 2>&1 | tail -3; fi
for cfg in "$@"; do
  envs=(); args=()
  for w in $cfg; do if [[ $w == [A-Z_]*=* && ${#args[@]} -eq 0 ]]; then envs+=("$w"); else args+=("$w"); fi; done
  env "${envs[@]}" python3 bench.py --no-cpu --nobj $NOBJ --steps $STEPS --warmup 1 "${args[@]}" 2>gpurun_out/quick.err | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}
print('%-48s %.4g %s  %.2f ms/step  frac %s  %s' % ('''$cfg''' or '(default)', d['value'], d['unit'], d['ms_per_step'],
      ('%.3f' % r['frac']) if r.get('frac') else '-', (d.get('config') or {}).get('kernel_form', '')))" || tail -3 gpurun_out/quick.err
done
