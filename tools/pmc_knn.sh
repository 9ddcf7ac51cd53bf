#!/bin/bash
# PMC passes of the k-NN search (kernel-trace only), per-launch sums
export TMPDIR=/tmp
TAG=${1:-knn}
ARGS="--workload knn --nobj ${NOBJ:-100000} --steps 1 --warmup 1 --no-cpu"
SETS=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
  "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT"
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_SMEM"
)
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_${TAG}_$i -- python3 bench.py $ARGS > gpurun_out/pmc_${TAG}_$i.log 2>&1
done
python3 - "$TAG" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
for d in sorted(glob.glob('gpurun_out/pmc_%s_*/' % tag)):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'knn' not in k: continue
            k = k.split('(')[0].replace('void fz::', '')[:50] + ' grid=' + r.get('Grid_Size', '?')
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
        for k in agg:
            for c in agg[k]:
                print('%-60s %-28s per_launch=%.6g n=%d' % (k, c, agg[k][c] / cnt[(k, c)], cnt[(k, c)]))
PY
