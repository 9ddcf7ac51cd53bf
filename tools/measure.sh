#!/bin/bash
# Full measurement set for profiles/ (run through gpurun; writes gpurun_out/m_*).  Release build only.
#   ./tools/measure.sh
export TMPDIR=/tmp
export FZ_BENCH_NO_EXTRA=1
O=gpurun_out
mkdir -p $O
FZ_BENCH_NO_EXTRA= python3 bench.py > $O/m_bench_fit_predict_modeA.json 2> $O/m_bench_fit_predict_modeA.err
python3 bench.py --exact --no-cpu > $O/m_bench_fit_predict_modeA_exact_evidence.json 2>/dev/null
FZ_HIST=0 python3 bench.py --no-cpu > $O/m_bench_fit_predict_modeA_k_fused_round2_kernel.json 2>/dev/null
python3 bench.py --noise-scale 3 --nobj 262144 --no-cpu > $O/m_bench_fit_predict_noise3.json 2>/dev/null
python3 bench.py --noise-scale 10 --nobj 262144 --no-cpu > $O/m_bench_fit_predict_noise10.json 2>/dev/null
python3 bench.py --model-err varying --no-cpu > $O/m_bench_fit_predict_modeA_varying_model_errors.json 2>/dev/null
python3 bench.py --mode B --no-cpu > $O/m_bench_fit_predict_modeB.json 2>/dev/null
python3 bench.py --mode Ai --no-cpu > $O/m_bench_fit_predict_modeAi.json 2>/dev/null
python3 bench.py --mode A --mask-frac 0.02 --no-cpu > $O/m_bench_fit_predict_masked.json 2>/dev/null
python3 bench.py --mode A --mask-frac 0.2 --no-cpu > $O/m_bench_fit_predict_masked_20pct.json 2>/dev/null
python3 bench.py --mode B --mask-frac 0.02 --no-cpu > $O/m_bench_fit_predict_masked_modeB.json 2>/dev/null
python3 bench.py --mode An --no-cpu > $O/m_bench_fit_predict_modeAn_no_dim_prior.json 2>/dev/null
python3 bench.py --mode Bn --no-cpu > $O/m_bench_fit_predict_modeBn_no_dim_prior.json 2>/dev/null
python3 bench.py --mode A --prior 64 --no-cpu > $O/m_bench_fit_predict_prior.json 2>/dev/null
python3 bench.py --workload fit --nobj 100000 --nmodel 10000 --no-cpu --steps 5 > $O/m_bench_fit_planes.json 2>/dev/null
python3 bench.py --workload predict --nobj 100000 --nmodel 10000 --no-cpu --steps 5 > $O/m_bench_predict_planes.json 2>/dev/null
python3 bench.py --workload predict --nobj 20000 --nmodel 100000 --no-cpu --steps 5 > $O/m_bench_predict_planes_1e5_models.json 2>/dev/null
python3 bench.py --kde grid --nobj 262144 --no-cpu --steps 2 > $O/m_bench_fit_predict_grid_kde.json 2>/dev/null
python3 bench.py --label-err varying --nobj 262144 --no-cpu --steps 2 > $O/m_bench_fit_predict_many_kernel_widths.json 2>/dev/null
python3 bench.py --mode C --model-err varying --nobj 20000 --nmodel 10000 --no-cpu --steps 2 > $O/m_bench_fit_predict_modeC.json 2>/dev/null
python3 bench.py --workload knn --nobj 100000 --no-cpu > $O/m_bench_knn.json 2>/dev/null
python3 bench.py --workload summarize --nobj 1000000 --no-cpu > $O/m_bench_summarize.json 2>/dev/null
for nb in 4 6 7 8 12 16 24 32; do python3 bench.py --nband $nb --nobj 262144 --no-cpu --steps 2 > $O/m_bench_fit_predict_${nb}bands.json 2>/dev/null; done
python3 bench.py --nband 16 --mode B --nobj 262144 --no-cpu --steps 2 > $O/m_bench_fit_predict_16bands_modeB.json 2>/dev/null
python3 bench.py --nband 16 --model-err varying --nobj 262144 --no-cpu --steps 2 > $O/m_bench_fit_predict_16bands_varying_model_errors.json 2>/dev/null
python3 bench.py --nband 32 --mode B --nobj 262144 --no-cpu --steps 2 > $O/m_bench_fit_predict_32bands_modeB.json 2>/dev/null
python3 bench.py --nband 12 --mode B --nobj 262144 --no-cpu --steps 2 > $O/m_bench_fit_predict_12bands_modeB.json 2>/dev/null
for nb in 7 8; do python3 bench.py --nband $nb --mode B --nobj 262144 --no-cpu --steps 2 > $O/m_bench_fit_predict_${nb}bands_modeB.json 2>/dev/null; done
for nb in 6 7 8; do python3 bench.py --nband $nb --model-err varying --nobj 262144 --no-cpu --steps 2 > $O/m_bench_fit_predict_${nb}bands_varying_model_errors.json 2>/dev/null; done
python3 bench.py --workload predict --nobj 200000 --nmodel 5000 --no-cpu --steps 5 > $O/m_bench_predict_planes_5000_models.json 2>/dev/null
python3 bench.py --workload predict --nobj 50000 --nmodel 20000 --no-cpu --steps 5 > $O/m_bench_predict_planes_20000_models.json 2>/dev/null
FZ_PLANE_ROWS=0 python3 bench.py --workload predict --nobj 100000 --nmodel 10000 --no-cpu --steps 5 > $O/m_bench_predict_planes_k_plane_fused.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/m_stats_headline -- python3 bench.py --no-cpu --steps 3 --warmup 1 > $O/m_stats_headline.log 2>&1
FZ_BENCH_NO_EXTRA= rocprofv3 --kernel-trace --stats --output-format csv -d $O/m_stats_default_line -- python3 bench.py --no-cpu > $O/m_stats_default_line.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/m_stats_summarize -- python3 bench.py --workload summarize --nobj 1000000 --no-cpu --steps 3 --warmup 1 > $O/m_stats_summarize.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/m_stats_planes -- python3 bench.py --workload fit --nobj 100000 --nmodel 10000 --no-cpu --steps 5 --warmup 1 > $O/m_stats_planes.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/m_stats_predict -- python3 bench.py --workload predict --nobj 100000 --nmodel 10000 --no-cpu --steps 5 --warmup 1 > $O/m_stats_predict.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/m_stats_bands16 -- python3 bench.py --nband 16 --nobj 262144 --no-cpu --steps 3 --warmup 1 > $O/m_stats_bands16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/m_stats_grid -- python3 bench.py --kde grid --nobj 262144 --no-cpu --steps 3 --warmup 1 > $O/m_stats_grid.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/m_stats_masked -- python3 bench.py --mask-frac 0.2 --no-cpu --steps 3 --warmup 1 > $O/m_stats_masked.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/m_stats_modec -- python3 bench.py --mode C --model-err varying --nobj 20000 --nmodel 10000 --no-cpu --steps 3 --warmup 1 > $O/m_stats_modec.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/m_stats_knn -- python3 bench.py --workload knn --nobj 100000 --no-cpu --steps 3 --warmup 1 > $O/m_stats_knn.log 2>&1
tail -c 600 $O/m_bench_fit_predict_modeA.json
