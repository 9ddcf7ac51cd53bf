#!/bin/bash
# one bench line per likelihood mode (262144 objects x 1e5 models); pass "test" to run the parity suites first
export TMPDIR=/tmp
export FZ_BENCH_NO_EXTRA=1
O=gpurun_out
mkdir -p $O
if [ "$1" == "test" ]; then
  timeout -k 10 1000 python -m pytest tests/test_hip_parity.py tests/test_hip_fuzz.py tests/test_hip_fullsize.py tests/test_hip_prior.py -m gpu -x -q > $O/modes_t.log 2>&1 || { tail -40 $O/modes_t.log; exit 1; }
  tail -2 $O/modes_t.log
fi
run() { python3 bench.py --no-cpu --nobj 262144 --steps 2 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g evals/s  %.1f ms' % (d['value'], d['ms_per_step']))"; }
for cfg in "" "--model-err varying" "--mode B" "--mode Ai" "--mode B --mask-frac 0.02"; do
  echo "$cfg | $(run $cfg)"
done
