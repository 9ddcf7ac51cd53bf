#!/bin/bash
# list-free form (FZ_NOLIST=1): parity suites under it, then headline / broad-likelihood lines with and without
export TMPDIR=/tmp
export FZ_BENCH_NO_EXTRA=1
O=gpurun_out
mkdir -p $O
if [ "$1" == "test" ]; then
  FZ_NOLIST=1 timeout -k 10 1000 python -m pytest tests/test_hip_parity.py tests/test_hip_fuzz.py tests/test_hip_fullsize.py -m gpu -q > $O/nl_t.log 2>&1 || { grep -E "^FAILED|^ERROR" $O/nl_t.log | head -30; tail -3 $O/nl_t.log; }
  tail -2 $O/nl_t.log
fi
run() { python3 bench.py --no-cpu --nobj 262144 --steps 2 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g evals/s  %.1f ms' % (d['value'], d['ms_per_step']))"; }
for cfg in "" "--noise-scale 3" "--noise-scale 10" "--noise-scale 2"; do
  echo "$cfg | auto $(run $cfg) | lists $(FZ_NOLIST=0 run $cfg) | no lists $(FZ_NOLIST=1 run $cfg)"
done
