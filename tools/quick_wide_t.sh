#!/bin/bash
BANDS="6 7 8" EXTRA="--model-err varying" ./tools/quick_wide.sh
python3 -m pytest tests/test_hip_parity.py -m gpu -x -q --tb=short 2>&1 | tail -3
