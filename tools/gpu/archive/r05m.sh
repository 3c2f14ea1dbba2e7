#!/bin/bash
# round 5, call m: rolling-operand variant of k_conv_wino8 (CSM_WINO_ROLL=1): correctness and timing against the shipped form
O=gpurun_out/r05m; mkdir -p $O
export CSM_SYNTHETIC_WEIGHTS=1
CSM_WINO_ROLL=1 timeout 600 python -m pytest tests/test_gpu_winograd.py -x -q -k "not kernel_form" > $O/pytest_roll.txt 2>&1; tail -3 $O/pytest_roll.txt
CSM_WINO_ROLL=1 timeout 600 python tools/wino_debug.py bench > $O/bench_roll.txt 2>&1; grep -v amdgpu.ids $O/bench_roll.txt
