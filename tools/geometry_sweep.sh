for args in "" "--model-err varying" "--mode B" "--mode Ai" "--mode An" "--mode Bn" "--prior 64"; do
  for cfg in default 2,16 4,8 2,8; do
    if [ $cfg = default ]; then unset FZ_FUSED_CFG; else export FZ_FUSED_CFG=$cfg; fi
    r=$(timeout -k 10 200 python bench.py --no-cpu $args --nobj 262144 --steps 3 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*")
    echo "[$args] cfg=$cfg $r"
  done
done
export FZ_NO_SPLIT=1
for args in "--mask-frac 0.05" "--mask-frac 0.05 --mode B" "--mask-frac 0.05 --mode Ai"; do
  for cfg in default 2,16 4,8 2,8; do
    if [ $cfg = default ]; then unset FZ_FUSED_CFG; else export FZ_FUSED_CFG=$cfg; fi
    r=$(timeout -k 10 200 python bench.py --no-cpu $args --nobj 262144 --steps 3 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*")
    echo "[masked $args] cfg=$cfg $r"
  done
done
