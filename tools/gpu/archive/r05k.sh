#!/bin/bash
O=gpurun_out/r05k; mkdir -p $O
export CSM_SYNTHETIC_WEIGHTS=1
timeout 1500 python -m pytest tests/test_gpu_winograd.py -x -q --durations=6 > $O/pytest.txt 2>&1; tail -14 $O/pytest.txt
