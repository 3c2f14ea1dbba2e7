#!/bin/bash
# round 4, GPU call A: interleaved DMA issue (CSM_ILV=1, the default build) against the burst form (libcsm355_noilv.so), bit-exactness first
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04a; mkdir -p $O
NOILV=$PWD/cartoonsegmentation_amd/libcsm355_noilv.so
timeout 1200 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "tile_configurations or repeated_runs or conv_bit_exact" > $O/pytest_nets.log 2>&1
tail -3 $O/pytest_nets.log
CF="6 7 8 9 11 12 38 39 40 41 42 43 44 45 47 48"
CFGS="$CF" timeout 900 python tools/conv_bench8.py > $O/cb8_ilv.txt 2>&1
CSM_LIB=$NOILV CFGS="$CF" timeout 900 python tools/conv_bench8.py > $O/cb8_noilv.txt 2>&1
tail -1 $O/cb8_ilv.txt $O/cb8_noilv.txt
LP_BATCH=8 timeout 900 python tools/layer_profile.py > $O/lp8_ilv.txt 2>&1
CSM_LIB=$NOILV LP_BATCH=8 timeout 900 python tools/layer_profile.py > $O/lp8_noilv.txt 2>&1
grep "^==" $O/lp8_ilv.txt $O/lp8_noilv.txt
