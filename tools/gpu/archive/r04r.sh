#!/bin/bash
# round 4, GPU call R: ZoeDepth with both TTA passes in one core run: parity tests, then the zoe frame variant
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dpt_beit.py tests/test_gpu_nets.py -x -q -m gpu -k "zoe" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 600 python - > $O/zoe_variant.txt 2>&1 <<'PY'
import json, torch, bench
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
wl = bench.FrameWorkload(1024, 0, dev, batch=1)
print(json.dumps(wl._zoe_variant()))
PY
grep -v amdgpu.ids $O/zoe_variant.txt | cut -c1-900
