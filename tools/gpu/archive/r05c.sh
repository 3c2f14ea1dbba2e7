#!/bin/bash
# round 5, call c: k_conv_wino with the compile-time-indexed epilogue (the run-time r loop was a 50 K-cycle waterfall per block)
O=gpurun_out/r05c; mkdir -p $O
export CSM_SYNTHETIC_WEIGHTS=1
timeout 300 python tools/wino_debug.py check > $O/wino_check.txt 2>&1; tail -8 $O/wino_check.txt | grep -v amdgpu.ids
timeout 600 python tools/wino_debug.py bench > $O/wino_bench.txt 2>&1; grep -v amdgpu.ids $O/wino_bench.txt
timeout 600 python -m pytest tests/test_gpu_winograd.py -x -q > $O/pytest_wino.txt 2>&1; tail -5 $O/pytest_wino.txt
timeout 900 python -m pytest tests/test_gpu_zoe_fullsize.py -q -k pipeline > $O/pytest_zoe.txt 2>&1; tail -25 $O/pytest_zoe.txt
