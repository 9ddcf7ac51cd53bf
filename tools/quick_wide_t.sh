#!/bin/bash
python3 -m pytest tests/test_hip_parity.py -m gpu -x -q -k "wide" --tb=short 2>&1 | tail -5
BANDS="24 32 8" ./tools/quick_wide.sh
BANDS="32" EXTRA="--mode B" ./tools/quick_wide.sh
BANDS="32" EXTRA="--mode B" FZ_HIST_NOSCRB=1 ./tools/quick_wide.sh
