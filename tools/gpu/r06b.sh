#!/bin/bash
# round 6, call b: per-layer profile at batch 8 (which small-map 3x3 layers carry the direct chain's time), write-burst ablation of k_conv_wino4
R=$PWD; O=$R/gpurun_out/r06b; mkdir -p $O
export CSM_SYNTHETIC_WEIGHTS=1
LP_BATCH=8 LP_TOP=60 timeout 600 python tools/layer_profile.py > $O/layer_profile_b8.txt 2>&1; tail -5 $O/layer_profile_b8.txt
CSM_LIB=$R/cartoonsegmentation_amd/libcsm355_dev.so WINO4_VARIANTS=0,8,64,0,8,64 timeout 300 python tools/wino4_debug.py variants 2>&1 | grep -v amdgpu | tee $O/wino4_write_burst.txt
