#!/bin/bash
# GPU tests + the predict bench line (k_plane_fused) + the 2-rank rehearsal
export TMPDIR=/tmp
O=gpurun_out/r3check; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/gputest.log
python3 bench.py --workload predict --nobj 100000 --nmodel 10000 --no-cpu --steps 5 > $O/predict.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('$O/predict.json').read().strip().splitlines()[-1]); print('predict %.3f ms/step, %.0f GB/s, %s' % (d['ms_per_step'], d['roofline']['achieved'], d['roofline']['kernel']))"
FZ_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --nobj 200000 --no-cpu > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "2rank rc=$?"
tail -c 1800 $O/bench_2rank_gloo.json
