#!/bin/bash
# round 4, GPU call W: plain DMA kernel with one barrier per two chunks (cfg 53 = 64 x 64, 54 = 128 x 32): bit-exactness, then the 1x1 / short-K layers
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04w; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "tile_configurations or repeated_runs" > $O/pytest_nets.log 2>&1
tail -3 $O/pytest_nets.log
CFGS="6 53 12 54" ONLY="1 5 6 7 4 23 24" timeout 600 python tools/conv_bench8.py > $O/pair.txt 2>&1
cut -c1-150 $O/pair.txt | grep -v amdgpu.ids
