#!/bin/bash
# final regression: every GPU test, the smoke entry point, the default bench line
export TMPDIR=/tmp
O=gpurun_out/r3final; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/gputest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python3 bench.py --steps 5 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python3 -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('value %.4g %s  ms/step %.2f  frac %.3f  traffic %s  cpu %.3g' % (d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value']))
for k in ('roofline_fp64','roofline_general','roofline_modeB'): print(k, '%.4g' % d[k]['value'], '%.3f' % d[k]['frac'])"
