#!/bin/bash
# PMC passes for the bench kernels (separate runs, kernel-trace only; never combined with sys/hip traces) at the bench's own
# launch shape (default 1e6 objects x 1e5 models, one launch per step), per-kernel per-launch sums, and the HBM-traffic entries
# of profiles/pmc_latest.json (FETCH_SIZE x 2 on gfx950 for wide coalesced reads, MI355X_MICROARCH.md, + WRITE_SIZE; both in KB).
#   BENCH_ARGS="--mode B" TAG=modeB ./tools/pmc.sh      -> gpurun_out/pmc_<TAG>.txt, gpurun_out/pmc_<TAG>.json
export TMPDIR=/tmp
export FZ_BENCH_NO_EXTRA=1
TAG=${TAG:-headline}
ARGS="--nobj ${NOBJ:-1000000} --nmodel ${NMODEL:-100000} --steps 1 --warmup 1 --no-cpu ${BENCH_ARGS}"
SETS=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
  "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
  "SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_LDS_ATOMIC"
  "FETCH_SIZE"
  "WRITE_SIZE"
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
)
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_${TAG}_$i -- python3 bench.py $ARGS > gpurun_out/pmc_${TAG}_$i.log 2>&1
done
python3 - "$TAG" "$ARGS" <<'PY' | tee gpurun_out/pmc_${TAG}.txt
import csv, glob, collections, sys, json
tag, args = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(dict)
for d in sorted(glob.glob('gpurun_out/pmc_%s_*/' % tag)):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'fz::' not in k: continue
            k = k.split('(')[0].replace('void fz::', '')[:70]
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
        for k in agg:
            for c in agg[k]:
                v = agg[k][c] / cnt[(k, c)]
                tot[k][c] = v
                print('%-72s %-26s per_launch=%.6g n=%d' % (k, c, v, cnt[(k, c)]))
out = {}
for k, c in tot.items():
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        out[k] = {'fetch_KB': c['FETCH_SIZE'], 'write_KB': c['WRITE_SIZE'], 'hbm_bytes_per_launch': (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024.0}
sys.path.insert(0, '.')
from frankenz_amd._lib import source_id
json.dump({'bench_args': args, 'source_id': source_id(), 'kernels': out}, open('gpurun_out/pmc_%s.json' % tag, 'w'), indent=1)
PY
