#!/bin/bash
export FZ_BENCH_NO_EXTRA=1
export TMPDIR=/tmp
O=gpurun_out
python3 bench.py --mode A --mask-frac 0.02 --no-cpu > $O/m_bench_fit_predict_masked.json 2>/dev/null
python3 bench.py --mode A --mask-frac 0.2 --no-cpu > $O/m_bench_fit_predict_masked_20pct.json 2>/dev/null
python3 bench.py --mode B --mask-frac 0.02 --no-cpu > $O/m_bench_fit_predict_masked_modeB.json 2>/dev/null
FZ_HIST_OBJMASK=0 python3 bench.py --mode A --mask-frac 0.2 --no-cpu > $O/m_bench_fit_predict_masked_20pct_split_launches.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/m_stats_masked -- python3 bench.py --mask-frac 0.2 --no-cpu --steps 3 --warmup 1 > $O/m_stats_masked.log 2>&1
for f in masked masked_20pct masked_modeB masked_20pct_split_launches; do python3 -c "
import json; d=json.loads(open('$O/m_bench_fit_predict_$f.json').read().strip().splitlines()[-1]); print('$f %.4g evals/s %.1f ms %s' % (d['value'], d['ms_per_step'], d['config']['kernel_form']))"; done
