#!/bin/bash
# round-3 regression set: GPU tests, the default bench line, a 2-rank gloo rehearsal of the N > 1 bench path on one GPU
export TMPDIR=/tmp
O=gpurun_out/r3check; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" | tee $O/status.txt; tail -3 $O/gputest.log
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/status.txt
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r3check/bench_default.json').read().strip().splitlines()[-1])
print('headline %.4g evals/s %.2f ms frac %.3f form %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['kernel_form']))
for k in ('roofline_fp64', 'roofline_general', 'roofline_modeB'):
    print(k, '%.4g evals/s frac %.3f %s' % (d[k]['value'], d[k]['frac'], d[k]['kernel']))
print('cpu', d['cpu_baseline']['value'], d['speedup_vs_cpu_core'])
PY
FZ_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --nobj 200000 --no-cpu > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "2rank rc=$?" | tee -a $O/status.txt
tail -c 1500 $O/bench_2rank_gloo.json
