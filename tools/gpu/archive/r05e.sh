#!/bin/bash
# round 5, call e: the eight-wave form k_conv_wino8 (two waves per SIMD): correctness, timing next to the four-wave form, its ablations
R=$PWD; O=$R/gpurun_out/r05e; mkdir -p $O
export CSM_SYNTHETIC_WEIGHTS=1
timeout 300 python tools/wino_debug.py check > $O/wino_check8.txt 2>&1; tail -8 $O/wino_check8.txt | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_winograd.py -x -q > $O/pytest_wino8.txt 2>&1; tail -5 $O/pytest_wino8.txt
CSM_WINO_WAVES=4 timeout 600 python -m pytest tests/test_gpu_winograd.py -x -q > $O/pytest_wino4.txt 2>&1; tail -3 $O/pytest_wino4.txt
timeout 600 python tools/wino_debug.py bench > $O/wino_bench8.txt 2>&1; grep -v amdgpu.ids $O/wino_bench8.txt
CSM_LIB=$R/cartoonsegmentation_amd/libcsm355_dev.so timeout 600 python tools/wino_debug.py variants > $O/variants8.txt 2>&1; grep -v amdgpu.ids $O/variants8.txt
