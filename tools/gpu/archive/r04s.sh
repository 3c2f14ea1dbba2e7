#!/bin/bash
# round 4, GPU call S: attention with the relative-position bias window in LDS (DMA): parity, then the core profile
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dpt_beit.py -x -q -m gpu > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 300 python tools/zoe_core_profile.py 672 672 > $O/zoe_core.txt 2>&1
timeout 300 python tools/zoe_core_profile.py 384 512 >> $O/zoe_core.txt 2>&1
grep -v amdgpu.ids $O/zoe_core.txt
