rm -rf gpurun_out/pmc_*
TAG=headline bash tools/pmc.sh > /dev/null 2>&1
TAG=general BENCH_ARGS="--model-err varying" bash tools/pmc.sh > /dev/null 2>&1
TAG=modeB BENCH_ARGS="--mode B" bash tools/pmc.sh > /dev/null 2>&1
TAG=modeAi BENCH_ARGS="--mode Ai" bash tools/pmc.sh > /dev/null 2>&1
bash tools/pmc_knn.sh knn > gpurun_out/pmc_knn.txt 2>&1
bash tools/pmc_predict.sh > /dev/null 2>&1
TAG=modec NOBJ=20000 NMODEL=10000 BENCH_ARGS="--mode C --model-err varying" bash tools/pmc.sh > /dev/null 2>&1
ls gpurun_out/pmc_*.txt gpurun_out/pmc_*.json
grep -c per_launch gpurun_out/pmc_headline.txt gpurun_out/pmc_knn.txt
