export TMPDIR=/tmp
python3 bench.py --workload knn --nobj 100000 --steps 3 --warmup 1 --no-cpu > /tmp/k.json 2>/tmp/k.err; echo "knn rc=$?"; cut -c1-260 /tmp/k.json; tail -3 /tmp/k.err
FZ_KNN_NOBOX=1 python3 bench.py --workload knn --nobj 100000 --steps 2 --warmup 1 --no-cpu > /tmp/k2.json 2>/tmp/k2.err; echo "knn nobox rc=$?"; cut -c1-260 /tmp/k2.json
python -m pytest tests/test_hip_fullsize.py -m gpu -x -q -k "config4_knn_at_its_stated_shape" 2>&1 | tail -5
