#!/bin/bash
# PMC passes for the predict-from-plane kernels (separate runs, kernel-trace only).
#   ./tools_pmc_predict.sh            (NOBJ / NMODEL override the 1e5 x 1e4 plane)
export TMPDIR=/tmp
export FZ_BENCH_NO_EXTRA=1
ARGS="--workload predict --nobj ${NOBJ:-100000} --nmodel ${NMODEL:-10000} --steps 1 --warmup 1 --no-cpu"
SETS=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
  "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
  "FETCH_SIZE"
  "WRITE_SIZE"
  "GRBM_GUI_ACTIVE"
)
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcp_$i -- python3 bench.py $ARGS > gpurun_out/pmcp_$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmcp_*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'k_plane_fused' not in k and 'k_stats' not in k and 'k_kde' not in k: continue
            k = k.split('(')[0].replace('void fz::', '')[:60]
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
        for k in agg:
            for c in agg[k]:
                print('%-42s %-26s per_launch=%.6g n=%d' % (k, c, agg[k][c] / cnt[(k, c)], cnt[(k, c)]))
PY
