#!/bin/bash
# launch geometry by band count (FZ_FUSED_CFG), 262 144 x 1e5 (131 072 from 12 bands), ms per launch
for nb in 4 6 7 8 12; do
  n=262144; if [ $nb -ge 12 ]; then n=131072; fi
  for args in "" "--model-err varying" "--mode B"; do
    for cfg in default 2,16 4,8 2,8; do
      if [ $cfg = default ]; then unset FZ_FUSED_CFG; else export FZ_FUSED_CFG=$cfg; fi
      r=$(timeout -k 10 200 python bench.py --no-cpu $args --nband $nb --nobj $n --steps 2 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*")
      echo "nband=$nb [$args] cfg=$cfg $r"
    done
  done
done
