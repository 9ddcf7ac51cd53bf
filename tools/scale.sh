#!/bin/bash
# The multi-GPU scaling curve in one pass, diagnosable from its output: bench.py through torch.distributed.run (RCCL) for every rank
# count in NS, per-N value / ms_compute / ms_gather_exposed / ms_fence / ranks seen, collected in gpurun_out/scale.json, and the check
# that the N = 1 line of this script equals the default bench line within 3 %.
#   tools/scale.sh                      # NS="1 2 4 8" on an 8-GPU node
#   NS="1 2" BACKEND=gloo tools/scale.sh   # rehearsal on a one-GPU box: ranks share the card, gathers through host memory (plumbing only)
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
NS=${NS:-"1 2 4 8"}
STEPS=${STEPS:-5}; WARMUP=${WARMUP:-2}
[ -n "$BACKEND" ] && export FZ_BENCH_BACKEND=$BACKEND
O=gpurun_out/scale; mkdir -p $O
NGPU=$(python3 -c "import torch; print(torch.cuda.device_count())")
echo "GPUs visible: $NGPU  backend: ${BACKEND:-nccl}  rank counts: $NS"
PORT=29517
for n in $NS; do
  if [ "$n" -gt "$NGPU" ] && [ -z "$BACKEND" ]; then echo "N=$n: only $NGPU GPUs here (set BACKEND=gloo to rehearse)"; continue; fi
  if [ "$n" = 1 ]; then
    FZ_BENCH_NO_EXTRA=1 python3 bench.py --gpus 1 --steps $STEPS --warmup $WARMUP --no-cpu ${BENCH_ARGS} > $O/n1.json 2> $O/n1.err
  else
    PORT=$((PORT+1))
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $n \
      --steps $STEPS --warmup $WARMUP --no-cpu ${BENCH_ARGS} > $O/n$n.json 2> $O/n$n.err
  fi
  echo "N=$n rc=$?"
done
python3 - "$O" $NS <<'PY'
import json, sys, os
O, ns = sys.argv[1], [int(v) for v in sys.argv[2:]]
rows = []
for n in ns:
    f = os.path.join(O, 'n%d.json' % n)
    if not os.path.exists(f) or not os.path.getsize(f):
        rows.append({'n_gpus': n, 'error': open(os.path.join(O, 'n%d.err' % n)).read()[-400:] if os.path.exists(os.path.join(O, 'n%d.err' % n)) else 'not run'}); continue
    d = json.loads(open(f).read().strip().splitlines()[-1])
    g = d.get('gather') or {}
    rows.append({'n_gpus': d['n_gpus'], 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'ms_compute': d.get('ms_compute'),
                 'ms_gather_exposed': d.get('ms_gather_exposed'), 'rounds': g.get('rounds'), 'bytes_received_per_rank': g.get('bytes_received_per_rank'),
                 'collective': g.get('collective'), 'kernel_form': d['config'].get('kernel_form'), 'scaling': d.get('scaling')})
base = next((r for r in rows if r.get('n_gpus') == 1 and 'value' in r), None)
for r in rows:
    if base and 'value' in r:
        r['speedup_vs_1'] = r['value'] / base['value']; r['efficiency'] = r['speedup_vs_1'] / r['n_gpus']
out = {'rows': rows}
# the N = 1 line against the default bench line (BENCH_rNN.json or a bench_default.json of this checkout), within 3 %
ref = None
for cand in sorted([f for f in os.listdir('.') if f.startswith('BENCH_r') and f.endswith('.json')])[::-1]:
    try:
        ref = json.load(open(cand)).get('parsed', {}).get('value') or ref
    except Exception:
        pass
    if ref: out['reference_line'] = cand; break
if base and ref:
    out['n1_vs_default_line'] = base['value'] / ref
    out['n1_matches_default_line_within_3pct'] = abs(base['value'] / ref - 1) <= 0.03
json.dump(out, open(os.path.join(O, 'scale.json'), 'w'), indent=1)
for r in rows:
    print(r)
print({k: v for k, v in out.items() if k != 'rows'})
PY
