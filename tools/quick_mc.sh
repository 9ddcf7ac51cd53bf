#!/bin/bash
# many dictionary widths: parity tests that exercise the window / class-sorted stacks, then the bench line (MC on / off)
export TMPDIR=/tmp
export FZ_BENCH_NO_EXTRA=1
O=gpurun_out
mkdir -p $O
if [ "$1" == "test" ]; then
  timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fuzz.py tests/test_hip_fullsize.py -m gpu -x -q > $O/mc_t1.log 2>&1 || { tail -40 $O/mc_t1.log; exit 1; }
  tail -2 $O/mc_t1.log
fi
python3 bench.py --label-err varying --nobj 262144 --no-cpu --steps 2 > $O/mc_b1.json 2>/dev/null; cut -c1-400 $O/mc_b1.json
FZ_NO_MC=1 python3 bench.py --label-err varying --nobj 262144 --no-cpu --steps 2 > $O/mc_b0.json 2>/dev/null; cut -c1-400 $O/mc_b0.json
