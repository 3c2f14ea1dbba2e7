#!/usr/bin/env python3
"""time the lowered nets on the GPU: ms, TFLOP/s (fp32 MFMA roof 157.3)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cartoonsegmentation_amd.weights import SynthWeights
from cartoonsegmentation_amd.runtime import CompiledProgram
from cartoonsegmentation_amd import nets


def bench(name, prog, ext, iters=10):
    cp = CompiledProgram(prog, 'cuda')
    for _ in range(2):
        cp.run(*ext)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        cp.run(*ext)
    en.record(); en.synchronize()
    ms = st.elapsed_time(en) / iters
    print("%-28s %8.3f ms  %7.2f GFLOP  %6.1f TFLOP/s  ws %.0f MB  ops %d" % (name, ms, prog.flops / 1e9, prog.flops / ms / 1e9,
          prog.workspace_floats * 4 / 1e6, len(prog.ops)), flush=True)
    return ms


if __name__ == '__main__':
    which = sys.argv[1:] or ['isnet']
    dev = 'cuda'
    if 'isnet' in which:
        for (n, s) in ((1, 720), (4, 720), (1, 1024)):
            p = nets.build_isnet(SynthWeights('isnet.'), n, s, s)
            bench("isnet n=%d %dx%d" % (n, s, s), p, [torch.rand(n, 4, s, s, device=dev), torch.empty(n, 1, s, s, device=dev)])
    if 'leres' in which:
        for s in (640, 1024):
            p = nets.build_leres(SynthWeights('leres.'), 1, s, s)
            bench("leres %dx%d" % (s, s), p, [torch.randn(1, 3, s, s, device=dev), torch.empty(1, 1, s, s, device=dev)])
    if 'rtmdet' in which:
        for s in (640, 1024):
            p, _ = nets.build_rtmdet(SynthWeights('rtmdet.'), 1, s, s)
            bench("rtmdet-l %dx%d" % (s, s), p.prog, p.example_ext(dev))
