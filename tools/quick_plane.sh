#!/bin/bash
# predict-from-plane check: the bench line with the register-resident kernel and without (FZ_PLANE_ROWS=0), at several row lengths
for shape in "100000 10000" "200000 5000" "240000 4200" "50000 20000" "60000 16500"; do
  set -- $shape
  for e in 1 0; do
    FZ_PLANE_ROWS=$e python3 bench.py --workload predict --nobj $1 --nmodel $2 --steps 5 --warmup 2 --no-cpu 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('N=$1 M=$2 FZ_PLANE_ROWS=$e: %.3f ms/step, %.0f GB/s, %s' % (d['ms_per_step'], d['roofline']['achieved'], d['roofline']['kernel']))"
  done
done
