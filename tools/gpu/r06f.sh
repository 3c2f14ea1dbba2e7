#!/bin/bash
# round 6: first GPU run of the vector-pipe grouped kernel -- parity, then timing with both tile widths
mkdir -p gpurun_out/r06f
timeout 900 python -m pytest tests/test_gpu_grouped.py "tests/test_gpu_nets.py::test_conv_bit_exact" -x -q 2>&1 | tail -15 > gpurun_out/r06f/tests.txt
cat gpurun_out/r06f/tests.txt
timeout 600 python tools/grouped_bench.py > gpurun_out/r06f/bench_auto.txt 2>&1
CSM_GROUPED_PX=4 timeout 600 python tools/grouped_bench.py > gpurun_out/r06f/bench_px4.txt 2>&1
CSM_GROUPED_PX=5 timeout 600 python tools/grouped_bench.py > gpurun_out/r06f/bench_px5.txt 2>&1
tail -12 gpurun_out/r06f/bench_auto.txt; echo; tail -10 gpurun_out/r06f/bench_px4.txt; echo; tail -10 gpurun_out/r06f/bench_px5.txt
