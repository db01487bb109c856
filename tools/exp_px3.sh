#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "poolx" 2>&1 | tail -15
python tools/px_time.py
