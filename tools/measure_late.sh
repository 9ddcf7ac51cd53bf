#!/bin/bash
# bench lines of what changed after the r3_v2 set was taken (12 / 24-band instantiations, mode B at 7 / 8 bands, per-model errors at
# 6-8 bands, no dimensionality prior, other plane row lengths) -> gpurun_out/m_bench_*.json
export TMPDIR=/tmp
export FZ_BENCH_NO_EXTRA=1
O=gpurun_out
for nb in 12 24; do python3 bench.py --nband $nb --nobj 262144 --no-cpu --steps 2 > $O/m_bench_fit_predict_${nb}bands.json 2>/dev/null; done
python3 bench.py --nband 12 --mode B --nobj 262144 --no-cpu --steps 2 > $O/m_bench_fit_predict_12bands_modeB.json 2>/dev/null
for nb in 7 8; do python3 bench.py --nband $nb --mode B --nobj 262144 --no-cpu --steps 2 > $O/m_bench_fit_predict_${nb}bands_modeB.json 2>/dev/null; done
for nb in 6 7 8; do python3 bench.py --nband $nb --model-err varying --nobj 262144 --no-cpu --steps 2 > $O/m_bench_fit_predict_${nb}bands_varying_model_errors.json 2>/dev/null; done
python3 bench.py --mode An --no-cpu > $O/m_bench_fit_predict_modeAn_no_dim_prior.json 2>/dev/null
python3 bench.py --mode Bn --no-cpu > $O/m_bench_fit_predict_modeBn_no_dim_prior.json 2>/dev/null
python3 bench.py --workload predict --nobj 200000 --nmodel 5000 --no-cpu --steps 5 > $O/m_bench_predict_planes_5000_models.json 2>/dev/null
python3 bench.py --workload predict --nobj 50000 --nmodel 20000 --no-cpu --steps 5 > $O/m_bench_predict_planes_20000_models.json 2>/dev/null
python3 bench.py --mode C --model-err varying --nobj 20000 --nmodel 10000 --no-cpu --steps 2 > $O/m_bench_fit_predict_modeC.json 2>/dev/null
for f in $O/m_bench_fit_predict_{12,24}bands.json $O/m_bench_fit_predict_12bands_modeB.json $O/m_bench_fit_predict_{7,8}bands_modeB.json $O/m_bench_fit_predict_{6,7,8}bands_varying_model_errors.json $O/m_bench_fit_predict_mode{An,Bn}_no_dim_prior.json $O/m_bench_predict_planes_{5000,20000}_models.json $O/m_bench_fit_predict_modeC.json; do python3 -c "
import json, os; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('%-60s %.4g %s  %.2f ms  %s' % (os.path.basename('$f')[8:-5], d['value'], d['unit'][:8], d['ms_per_step'], d.get('config',{}).get('kernel_form') or r.get('kernel')))"; done
