#!/bin/bash
python3 -m pytest tests/test_hip_sharded.py -m gpu -x -q --tb=short 2>&1 | tail -6
