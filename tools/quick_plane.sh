#!/bin/bash
# predict-from-plane check: the bench line with the register-resident kernel and without (FZ_PLANE_ROWS=0); TESTS=1 adds the parity subset
set -e
mkdir -p gpurun_out
if [ -n "$TESTS" ]; then python3 -m pytest tests -m gpu -x -q -k "predict or pred or plane or modec or golden" --tb=short 2>&1 | tail -15; fi
for cfg in ${CFGS:-"8,10"}; do
  echo "== FZ_PLANE_ROWS_CFG=$cfg"; FZ_PLANE_ROWS_CFG=$cfg python3 bench.py --workload predict --nobj 100000 --nmodel 10000 --steps 5 --warmup 2 --no-cpu 2>/dev/null | tail -1
done
echo "== FZ_PLANE_ROWS=0"; FZ_PLANE_ROWS=0 python3 bench.py --workload predict --nobj 100000 --nmodel 10000 --steps 5 --warmup 2 --no-cpu 2>/dev/null | tail -1
