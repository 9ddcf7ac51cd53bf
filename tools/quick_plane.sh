#!/bin/bash
# predict-from-plane check: parity subset (TESTS=1), then the bench line per row length: register-resident rows with / without the
# compaction of the entries that matter (FZ_PLANE_ROWS_CMP=0), and k_plane_fused (FZ_PLANE_ROWS=0)
if [ -n "$TESTS" ]; then python3 -m pytest tests -m gpu -x -q -k "predict or plane or modec or rows" --tb=short 2>&1 | tail -4; fi
for shape in ${SHAPES:-"100000 10000" "200000 5000" "50000 20000"}; do
  set -- $shape
  for e in "FZ_PLANE_ROWS=1" "FZ_PLANE_ROWS_CMP=0" "FZ_PLANE_ROWS=0"; do
    env $e python3 bench.py --workload predict --nobj $1 --nmodel $2 --steps 5 --warmup 2 --no-cpu 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('N=$1 M=$2 $e: %.3f ms/step, %.0f GB/s, %s' % (d['ms_per_step'], d['roofline']['achieved'], d['roofline']['kernel']))"
  done
done
