#!/bin/bash
# round 4, GPU call I: the weights-stationary 3x3 kernel (cfg 51 / 52) -- bit-exactness, then against the tuned tiles
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04i; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "tile_configurations or repeated_runs or conv_bit_exact" > $O/pytest_nets.log 2>&1
tail -5 $O/pytest_nets.log
CFGS="44 45 51 52" ONLY="11 18 25 26 27 28 29" timeout 900 python tools/conv_bench8.py > $O/cb8.txt 2>&1
cut -c1-140 $O/cb8.txt | tail -n 9
