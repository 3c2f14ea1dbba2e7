#!/usr/bin/env python3
"""csm_percentile_pair on 1024^2 planes (HIP events over 200 calls): a smooth depth ramp (the frame loop's case), uniform noise in
[200, 900) (every wave holds 64 different values) and two far-apart clusters (half of the plane misses a block's LDS window)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cartoonsegmentation_amd import _lib
from cartoonsegmentation_amd._lib import check, f64, i64, ptr, stream_ptr
L = _lib.load()
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:1024, 0:1024]
planes = {"ramp": (3.0 + 0.002 * yy + 0.0005 * xx + 0.3 * np.sin(xx / 90.0)).astype(np.float32),
          "uniform": rng.uniform(200, 900, (1024, 1024)).astype(np.float32),
          "clusters": np.where(rng.random((1024, 1024)) < 0.5, rng.normal(1, 0.01, (1024, 1024)), rng.normal(-5e4, 10, (1024, 1024))).astype(np.float32)}
sel = torch.zeros(L.csm_percentile_scratch_bytes(), dtype=torch.uint8, device='cuda')
out2 = torch.empty(2, device='cuda')
only = sys.argv[1:]
for name, a in planes.items():
    if only and name not in only:
        continue
    d = torch.from_numpy(a).cuda()
    for _ in range(5):
        check(L.csm_percentile_pair(ptr(d), i64(a.size), f64(2.0), f64(85.0), ptr(out2), ptr(sel), stream_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        check(L.csm_percentile_pair(ptr(d), i64(a.size), f64(2.0), f64(85.0), ptr(out2), ptr(sel), stream_ptr()))
    e1.record(); torch.cuda.synchronize()
    ok = np.array_equal(out2.cpu().numpy(), np.percentile(a.ravel(), [2.0, 85.0]).astype(np.float32))
    print("percentile_pair %-9s %.1f us per call, exact %s" % (name, e0.elapsed_time(e1) * 5.0, ok))
