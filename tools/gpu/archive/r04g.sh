#!/bin/bash
# round 4, GPU call G: three-stage interleaved-issue tiles (cfg 28..32) against the two-stage ones; the tuner may now pick them
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04g; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "tile_configurations or repeated_runs or conv_bit_exact or splitk" > $O/pytest_nets.log 2>&1
tail -3 $O/pytest_nets.log
CFGS="6 28 7 29 11 30 9 32" timeout 900 python tools/conv_bench8.py > $O/cb8.txt 2>&1
tail -n 1 $O/cb8.txt
LP_BATCH=8 timeout 900 python tools/layer_profile.py > $O/lp8.txt 2>&1
grep "^==" $O/lp8.txt
