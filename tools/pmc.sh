#!/bin/bash
# PMC passes for the bench kernels (separate runs, kernel-trace only; never combined with
# sys/hip traces).  Writes gpurun_out/pmc_<n>/ and prints per-kernel per-launch sums.
#   BENCH_ARGS="--mode B" ./tools/pmc.sh
export TMPDIR=/tmp
export FZ_BENCH_NO_EXTRA=1
ARGS="--nobj ${NOBJ:-262144} --nmodel 100000 --steps 1 --warmup 1 --no-cpu ${BENCH_ARGS}"
SETS=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
  "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
  "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU"
  "FETCH_SIZE"
  "WRITE_SIZE"
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
  "GRBM_GUI_ACTIVE GRBM_COUNT"
)
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_$i -- python3 bench.py $ARGS > gpurun_out/pmc_$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'fz::' not in k: continue
            k = k.split('(')[0].replace('void fz::', '')[:60]
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
        for k in agg:
            for c in agg[k]:
                print('%-42s %-26s per_launch=%.6g n=%d' % (k, c, agg[k][c] / cnt[(k, c)], cnt[(k, c)]))
PY
