#!/bin/bash
# round 5, call a: first contact of k_conv_wino with the GPU (correctness report + timing vs the tuned direct kernels), the new
# full-size zoe parity tests, median-3
O=gpurun_out/r05a; mkdir -p $O
export CSM_SYNTHETIC_WEIGHTS=1
timeout 300 python tools/wino_debug.py check > $O/wino_check.txt 2>&1; echo "check rc $?" >> $O/wino_check.txt
tail -40 $O/wino_check.txt
timeout 600 python tools/wino_debug.py bench > $O/wino_bench.txt 2>&1; echo "bench rc $?" >> $O/wino_bench.txt
cat $O/wino_bench.txt
timeout 600 python -m pytest tests/test_gpu_winograd.py -x -q > $O/pytest_wino.txt 2>&1; tail -15 $O/pytest_wino.txt
timeout 900 python -m pytest tests/test_gpu_zoe_fullsize.py -q --durations=5 > $O/pytest_zoe.txt 2>&1; tail -25 $O/pytest_zoe.txt
timeout 300 python -m pytest tests/test_gpu_warp.py -q -k "pointwise or filter or spatial" > $O/pytest_warp.txt 2>&1; tail -5 $O/pytest_warp.txt
