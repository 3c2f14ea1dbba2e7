#!/usr/bin/env python3
"""run N bench steps (8 frames each) and nothing else -- for `rocprofv3 --kernel-trace`: how much of a step is the GPU idle?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
wl = bench.FrameWorkload(1024, 0, torch.device('cuda'), 8)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
marker = torch.zeros(3, device='cuda') + 1            # a recognisable tiny kernel right before the measured steps
for _ in range(6):
    wl.step()
torch.cuda.synchronize()
print("6 steps: %.2f ms per step" % ((time.perf_counter() - t0) / 6 * 1e3), flush=True)
