#!/bin/bash
python3 -m pytest tests/test_hip_parity.py -m gpu -x -q -k "wide or masked or mask" --tb=short 2>&1 | tail -15
BANDS="12 16 24 32" ./tools/quick_wide.sh
BANDS="16 32" EXTRA="--mode B" ./tools/quick_wide.sh
BANDS="16" EXTRA="--model-err varying" ./tools/quick_wide.sh
BANDS="12" EXTRA="--mask-frac 0.02" ./tools/quick_wide.sh
