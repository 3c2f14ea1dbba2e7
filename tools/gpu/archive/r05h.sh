#!/bin/bash
# round 5, call h: 16 x 16-pixel block-tile geometry of k_conv_wino8 (GEO 1): correctness under both geometries, timing
O=gpurun_out/r05h; mkdir -p $O
export CSM_SYNTHETIC_WEIGHTS=1
CSM_WINO_GEO=1 timeout 300 python tools/wino_debug.py check > $O/check_geo1.txt 2>&1; tail -8 $O/check_geo1.txt | grep -v amdgpu.ids
CSM_WINO_GEO=1 timeout 600 python -m pytest tests/test_gpu_winograd.py -x -q > $O/pytest_geo1.txt 2>&1; tail -3 $O/pytest_geo1.txt
timeout 600 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_warp.py -x -q -k "winograd or multi_frame" > $O/pytest_auto.txt 2>&1; tail -3 $O/pytest_auto.txt
CSM_WINO_GEO=0 timeout 600 python tools/wino_debug.py bench > $O/bench_geo0.txt 2>&1; grep -v amdgpu.ids $O/bench_geo0.txt
timeout 600 python tools/wino_debug.py bench > $O/bench_auto.txt 2>&1; grep -v amdgpu.ids $O/bench_auto.txt
