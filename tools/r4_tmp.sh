export TMPDIR=/tmp; export FZ_BENCH_NO_EXTRA=1
for v in "" "FZ_HIST_NOSCRB=1"; do
env $v python3 bench.py --nobj 262144 --steps 3 --warmup 1 --no-cpu --mode B > /tmp/b.json 2>/tmp/b.err
python3 -c "
import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print('B $v value %.4g  ms/step %.2f  frac %.3f form %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['kernel_form']))"
done
