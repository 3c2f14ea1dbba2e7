#!/bin/bash
O=gpurun_out/r05l; mkdir -p $O
export CSM_SYNTHETIC_WEIGHTS=1
timeout 600 python -m pytest tests/test_gpu_warp.py -x -q -k "multi_frame" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/warp_multiframe.txt
import sys, torch, json
sys.path.insert(0, '.')
import bench
from cartoonsegmentation_amd import ops
wl = bench.FrameWorkload.__new__(bench.FrameWorkload)
wl.device = torch.device('cuda', 0); wl.ops = ops
r = bench.FrameWorkload._warp_points(wl)
for k in ('tiled_1024', 'tiled_1024_3streams', 'tiled_1024_multiframe', 'tiled_2048', 'tiled_2048_3streams', 'tiled_2048_multiframe'):
    print(k, r[k]['us_per_frame'], r[k]['frac_of_hbm_peak'])
PY
