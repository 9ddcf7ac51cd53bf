#!/bin/bash
# PMC passes for predict() from a stored plane (k_plane_rows / k_plane_fused): per-launch sums.  -> gpurun_out/pmc_predict.txt
export TMPDIR=/tmp
ARGS="--workload predict --nobj ${NOBJ:-100000} --nmodel ${NMODEL:-10000} --steps 2 --warmup 1 --no-cpu"
SETS=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
  "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH"
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU"
  "FETCH_SIZE"
  "WRITE_SIZE"
)
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_predict_$i -- python3 bench.py $ARGS > gpurun_out/pmc_predict_$i.log 2>&1
done
python3 - <<'PY' | tee gpurun_out/pmc_predict.txt
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_predict_*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'k_plane' not in k: continue
            k = k.split('(')[0].replace('void fz::', '')[:70]
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
        for k in agg:
            for c in agg[k]:
                print('%-40s %-26s per_launch=%.6g n=%d' % (k, c, agg[k][c] / cnt[(k, c)], cnt[(k, c)]))
PY
