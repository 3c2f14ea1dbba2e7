#!/usr/bin/env python3
"""do two layer programs launched on two HIP streams overlap at batch 1?  (RTMDet n=1 on one stream, LeReS n=1 on another)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CSM_SYNTHETIC_WEIGHTS"] = "1"
import torch
from cartoonsegmentation_amd.nets import build_leres, build_rtmdet
from cartoonsegmentation_amd.runtime import CompiledProgram
from cartoonsegmentation_amd.weights import SynthWeights
dev = torch.device('cuda')
N = int(os.environ.get('PROBE_BATCH', '1'))
def mk(prog):
    cp = CompiledProgram(prog, dev)
    ext = sorted((b for b in cp.prog.bufs if b.ext >= 0), key=lambda b: b.ext)
    ts = [torch.randn(b.n, b.c, b.h, b.w, device=dev) for b in ext]
    cp.run(*ts); torch.cuda.synchronize()
    return cp, ts
a, ta = mk(build_rtmdet(SynthWeights('rtmdet.'), N, 640, 640)[0].prog)
b, tb = mk(build_leres(SynthWeights('leres.'), N, 640, 640))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def wall(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def seq():
    a.run(*ta); b.run(*tb)
def par():
    with torch.cuda.stream(sa): a.run(*ta)
    with torch.cuda.stream(sb): b.run(*tb)
def only_a(): a.run(*ta)
def only_b(): b.run(*tb)
print("batch %d: rtmdet %.2f ms, leres %.2f ms, both on one stream %.2f ms, on two streams %.2f ms"
      % (N, wall(only_a), wall(only_b), wall(seq), wall(par)), flush=True)
