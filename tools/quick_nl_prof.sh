#!/bin/bash
export TMPDIR=/tmp
export FZ_BENCH_NO_EXTRA=1
O=gpurun_out
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/nl_stats -- python3 bench.py --noise-scale 3 --nobj 262144 --no-cpu --steps 2 --warmup 1 > $O/nl_stats.log 2>&1
f=$(ls -t $O/nl_stats/*/*kernel_stats.csv | head -1); head -6 $f | cut -c1-200
for set in "FETCH_SIZE" "WRITE_SIZE"; do
rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/nl_pmc_$set -- python3 bench.py --noise-scale 3 --nobj 262144 --no-cpu --steps 1 --warmup 1 > $O/nl_pmc_$set.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/nl_pmc_*/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_nl_' not in k: continue
        k = k.split('<')[0].replace('void fz::', '') + ' ' + r['Counter_Name']
        agg[k] += float(r['Counter_Value']); cnt[k] += 1
    for k in agg: print(k, 'per_launch=%.4g' % (agg[k] / cnt[k]), 'n=%d' % cnt[k])
PY
