#!/bin/bash
# scratch: GPU experiments of the moment (not part of the measurement set)
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -n 4 2>&1 | tail -2
python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -2
