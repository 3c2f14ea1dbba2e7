#!/bin/bash
# round 5, call n: persistent form k_conv_wino8p (CSM_WINO_PERSISTENT=1) on the rolling-operand pipeline: correctness and timing
O=gpurun_out/r05n; mkdir -p $O
export CSM_SYNTHETIC_WEIGHTS=1
CSM_WINO_PERSISTENT=1 timeout 600 python -m pytest tests/test_gpu_winograd.py -x -q -k "not kernel_form" > $O/pytest_p.txt 2>&1; tail -3 $O/pytest_p.txt
CSM_WINO_PERSISTENT=1 timeout 600 python tools/wino_debug.py check > $O/check_p.txt 2>&1; tail -3 $O/check_p.txt | grep -v amdgpu.ids
CSM_WINO_PERSISTENT=1 timeout 600 python tools/wino_debug.py bench > $O/bench_p.txt 2>&1; grep -v amdgpu.ids $O/bench_p.txt
