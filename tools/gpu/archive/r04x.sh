#!/bin/bash
# round 4, GPU call X: attention register allocation for 3 waves per SIMD on dense launches (two samples at 1765 tokens = 896 blocks): parity, A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04x; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dpt_beit.py -x -q -m gpu > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for opt in 2 0; do
  ATT_OPTIONS=$opt timeout 300 python tools/zoe_core_profile.py 672 672 2 2>&1 | grep -v amdgpu.ids | tee -a $O/zoe_core_n2.txt
done
