#!/bin/bash
# wide band sets: parity subset, then one line per band count / mode
python3 -m pytest tests/test_hip_parity.py -m gpu -x -q -k "wide or band or mask" --tb=short 2>&1 | tail -4
BANDS="9 12 16 20 24 32" ./tools/quick_wide.sh
BANDS="12 16 24 32" EXTRA="--mode B" ./tools/quick_wide.sh
BANDS="6 7 8 12 16" EXTRA="--model-err varying" ./tools/quick_wide.sh
BANDS="12 24" EXTRA="--mask-frac 0.02" ./tools/quick_wide.sh
