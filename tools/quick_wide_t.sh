#!/bin/bash
python3 -m pytest tests -m gpu -x -q -k "predict or plane or modec or rows" --tb=short 2>&1 | tail -4
