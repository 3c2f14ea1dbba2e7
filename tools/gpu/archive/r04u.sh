#!/bin/bash
# round 4, GPU call U: attention path test, then the headline bench line with the vendor GEMM context (no variants)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dpt_beit.py -x -q -m gpu -k "window or small" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 600 python bench.py --no-variants --no-cpu-baseline --no-iou > $O/bench.log 2>&1
grep '"metric"' $O/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['roofline']['frac'], json.dumps(d['roofline']['vendor_fp32_gemm_context']))"
