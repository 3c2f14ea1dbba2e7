#!/bin/bash
# round 4, GPU call J: attention kernel with the register prefetch of the next K / V tiles -- parity, then the core profile
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dpt_beit.py tests/test_gpu_kenburns.py -x -q -m gpu -k "dpt or beit or zoe or frame_lanes" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 300 python tools/zoe_core_profile.py 672 672 > $O/zoe_core.txt 2>&1
timeout 300 python tools/zoe_core_profile.py 384 512 >> $O/zoe_core.txt 2>&1
cat $O/zoe_core.txt
