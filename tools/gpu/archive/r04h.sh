#!/bin/bash
# round 4, GPU call H: compact fragment addressing in the persistent patch kernels (fewer VGPRs) + the 8-wave tile at two blocks per CU (cfg 50)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04h; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "tile_configurations or repeated_runs or conv_bit_exact" > $O/pytest_nets.log 2>&1
tail -3 $O/pytest_nets.log
CFGS="43 44 45 47 50 48" ONLY="0 2 3 10 11 12 13 14 15 16 17 19 20 21 22 25 28" timeout 900 python tools/conv_bench8.py > $O/cb8.txt 2>&1
tail -n 1 $O/cb8.txt
LP_BATCH=8 timeout 900 python tools/layer_profile.py > $O/lp8.txt 2>&1
grep "^==" $O/lp8.txt
