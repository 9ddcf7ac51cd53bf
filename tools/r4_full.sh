#!/bin/bash
# full regression of a build: every GPU test, smoke, the default bench line, k-NN line; PMC=1 adds the counter passes
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r4full}; mkdir -p $O
PYTHONUNBUFFERED=1 timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/gputest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python3 bench.py --steps 5 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python3 -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('value %.4g %s  ms/step %.2f  frac %.3f form %s cpu %.3g' % (d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d['config']['kernel_form'], d['cpu_baseline']['value']))
for k in ('roofline_general','roofline_modeB'): print(k, '%.4g' % d[k]['value'], '%.3f' % d[k]['frac'], d[k]['kernel'])"
python3 bench.py --workload knn --nobj 100000 --steps 3 --warmup 1 --no-cpu > $O/bench_knn.json 2> $O/bench_knn.err; echo "knn rc=$?"; cut -c1-250 $O/bench_knn.json
if [ -n "$PMC" ]; then
  bash tools/pmc_knn.sh r4knn > $O/pmc_knn.txt 2>&1; grep k_knn_mfma $O/pmc_knn.txt | head -40
  NOBJ=262144 bash tools/pmc_quick.sh r4hist > $O/pmc_hist.txt 2>&1; grep k_hist $O/pmc_hist.txt | head -40
fi
