#!/usr/bin/env python3
"""what the vendor library's fp32 GEMM (torch.mm -> hipBLASLt / rocBLAS, TF32 off) reaches on the GEMM shapes of the frame's conv layers:
a practical ceiling for exact-fp32 MFMA on this machine, next to this build's conv kernels (tools/conv_bench8.py)"""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
SHAPES = [("LeReS 1x1 1024->1024 @40^2 x8", 12800, 1024, 1024), ("LeReS 3x3 256->256 @160^2 x8 (as GEMM)", 204800, 256, 2304),
          ("LeReS 1x1 256->256 @160^2 x8", 204800, 256, 256), ("RTMDet 1x1 256->256 @40^2 x8", 12800, 256, 256),
          ("ISNet 3x3 64->64 @360^2 x16 (as GEMM)", 2073600, 64, 576), ("RTMDet 3x3 256->256 @80^2 x8 (as GEMM)", 51200, 256, 2304),
          ("big square", 8192, 8192, 8192)]
for name, M, N, K in SHAPES:
    a = torch.randn(M, K, device='cuda'); b = torch.randn(K, N, device='cuda')
    for _ in range(3):
        c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(5):
            c = a @ b
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    print("%-44s M %7d N %5d K %5d  %8.1f us  %6.1f TF/s" % (name, M, N, K, best * 1e3, 2.0 * M * N * K / best / 1e9), flush=True)
