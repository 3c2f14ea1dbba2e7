#!/bin/bash
# round 4, GPU call E: the DPT-BEiT core (small + BEiT-L 384x512 + ZoeDepth on it), the two zoe tests of test_gpu_nets, the 2-rank bench test
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04e; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_dpt_beit.py -x -q -m gpu --durations=8 > $O/pytest_dpt.log 2>&1; tail -15 $O/pytest_dpt.log
timeout 900 python -m pytest tests/test_gpu_nets.py tests/test_gpu_kenburns.py -x -q -m gpu -k "zoe or two_ranks" > $O/pytest_zoe.log 2>&1; tail -5 $O/pytest_zoe.log
