#!/bin/bash
# PMC passes for the direct grid KDE (--kde grid) and the many-width dictionary stack (--label-err varying), 262 144 objects
export TMPDIR=/tmp
export FZ_BENCH_NO_EXTRA=1
ARGS="--nobj 262144 --steps 1 --warmup 1 --no-cpu ${BENCH_ARGS:---kde grid}"
TAG=${TAG:-grid}
SETS=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
  "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
  "SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC"
)
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_${TAG}_$i -- python3 bench.py $ARGS > gpurun_out/pmc_${TAG}_$i.log 2>&1
done
python3 - "$TAG" <<'PY' | tee gpurun_out/pmc_${TAG}.txt
import csv, glob, collections, sys
tag = sys.argv[1]
for d in sorted(glob.glob('gpurun_out/pmc_%s_*/' % tag)):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'fz::k_fused' not in k and 'fz::k_hist' not in k: continue
            k = k.split('(')[0].replace('void fz::', '')[:60]
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
        for k in agg:
            for c in agg[k]:
                print('%-62s %-24s per_launch=%.5g n=%d' % (k, c, agg[k][c] / cnt[(k, c)], cnt[(k, c)]))
    for f in glob.glob(d + '**/*kernel_trace.csv', recursive=True)[:1]:
        if d.endswith('_1/'):
            t = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                t[r['Kernel_Name'].split('(')[0][:70]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
            for k, v in sorted(t.items(), key=lambda kv: -sum(kv[1]))[:6]: print('TIME %-70s n=%d avg=%.3f ms' % (k, len(v), sum(v) / len(v)))
PY
