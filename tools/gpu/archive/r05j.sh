#!/bin/bash
# round 5, call j: U stages issued first in k_conv_wino8's prologue; attention softmax (s - m) log2 e; checks + timings
O=gpurun_out/r05j; mkdir -p $O
export CSM_SYNTHETIC_WEIGHTS=1
timeout 900 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_dpt_beit.py tests/test_gpu_zoe_fullsize.py -x -q -k "not pipeline and not 672" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 600 python tools/wino_debug.py bench > $O/wino_bench.txt 2>&1; grep -v amdgpu.ids $O/wino_bench.txt
(timeout 200 python tools/zoe_core_profile.py 672 672) 2>/dev/null > $O/zoe_core.txt; cat $O/zoe_core.txt
