import os, sys, time
sys.path.insert(0, '/root/repo')
os.environ["CSM_SYNTHETIC_WEIGHTS"] = "1"
import torch
from cartoonsegmentation_amd.nets import build_leres, build_rtmdet
from cartoonsegmentation_amd.runtime import CompiledProgram
from cartoonsegmentation_amd.weights import SynthWeights
dev = torch.device('cuda')
def ev(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name, prog, shapes in (("leres n=8", build_leres(SynthWeights('leres.'), 8, 640, 640), [(8, 3, 640, 640), (8, 1, 640, 640)]),
                           ("rtmdet n=8", build_rtmdet(SynthWeights('rtmdet.'), 8, 640, 640)[0].prog, None)):
    cp = CompiledProgram(prog, dev)
    ext = sorted((b for b in cp.prog.bufs if b.ext >= 0), key=lambda b: b.ext)
    ts = [torch.randn(b.n, b.c, b.h, b.w, device=dev) for b in ext]
    cp.run(*ts); torch.cuda.synchronize()
    direct = ev(lambda: cp.run(*ts))
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        cp.run(*ts)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            cp.run(*ts)
    torch.cuda.synchronize()
    graph = ev(lambda: g.replay())
    print("%s: direct %.3f ms, hipGraph replay %.3f ms (%d ops)" % (name, direct, graph, len(cp.prog.ops)), flush=True)
