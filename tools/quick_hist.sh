#!/bin/bash
# k_hist vs k_fused: parity on a small problem, then one bench line per form / mode at 262 144 objects
export TMPDIR=/tmp
export FZ_BENCH_NO_EXTRA=1
O=gpurun_out/hist; mkdir -p $O; : > $O/results.txt
for cfg in "FZ_HIST_CFG=1,16" "FZ_EXACT_EVIDENCE=1"; do
  echo "== parity $cfg" | tee -a $O/results.txt
  env $cfg timeout -k 10 300 python3 tools/parity_quick.py 2>&1 | tail -8 | tee -a $O/results.txt
done
A="--nobj ${NOBJ:-262144} --no-cpu --steps 2 --warmup 1"
for cfg in "FZ_HIST_CFG=1,16" "FZ_EXACT_EVIDENCE=1"; do
  for extra in "" "--mode B" "--model-err varying" "--noise-scale 3"; do
    env $cfg timeout -k 10 300 python3 bench.py $A $extra 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg', '$extra', '%.4g evals/s  %.2f ms' % (d['value'], d['ms_per_step']), d['pdfs_normalised'])" | tee -a $O/results.txt
  done
done
