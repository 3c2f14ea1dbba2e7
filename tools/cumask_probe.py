#!/usr/bin/env python3
"""two layer programs on two HIP streams with disjoint CU masks (hipExtStreamCreateWithCUMask): does a static split of the chip
beat free-for-all sharing?  (a kernel on half the CUs sees twice as many tiles per CU: less grid quantisation)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CSM_SYNTHETIC_WEIGHTS"] = "1"
import torch
from cartoonsegmentation_amd.nets import build_isnet, build_leres, build_rtmdet
from cartoonsegmentation_amd.runtime import CompiledProgram
from cartoonsegmentation_amd.weights import SynthWeights
dev = torch.device('cuda')
N = int(os.environ.get('PROBE_BATCH', '8'))
hip = ctypes.CDLL('libamdhip64.so')
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if bits(32 * w + b)) for w in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(8), words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)
def mk(prog):
    cp = CompiledProgram(prog, dev)
    ext = sorted((b for b in cp.prog.bufs if b.ext >= 0), key=lambda b: b.ext)
    ts = [torch.randn(b.n, b.c, b.h, b.w, device=dev) for b in ext]
    cp.run(*ts); torch.cuda.synchronize()
    return cp, ts
a, ta = mk(build_rtmdet(SynthWeights('rtmdet.'), N, 640, 640)[0].prog)
i_, ti = mk(build_isnet(SynthWeights('isnet.'), 2 * N, 720, 720))
b, tb = mk(build_leres(SynthWeights('leres.'), N, 640, 640))
def wall(fn, n=4):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def seq():
    a.run(*ta); i_.run(*ti); b.run(*tb)
def two(sa, sb):
    def f():
        with torch.cuda.stream(sa): a.run(*ta); i_.run(*ti)
        with torch.cuda.stream(sb): b.run(*tb)
    return f
def one(s, cp, ts):
    def f():
        with torch.cuda.stream(s): cp.run(*ts)
    return f
print("batch %d: one stream (rtmdet + isnet + leres) %.2f ms" % (N, wall(seq)), flush=True)
print("two plain streams %.2f ms" % wall(two(torch.cuda.Stream(), torch.cuda.Stream())), flush=True)
print("rtmdet+isnet stream HIGH priority, leres normal %.2f ms" % wall(two(torch.cuda.Stream(priority=-1), torch.cuda.Stream())), flush=True)
print("leres stream HIGH priority, rtmdet+isnet normal %.2f ms" % wall(two(torch.cuda.Stream(), torch.cuda.Stream(priority=-1))), flush=True)
if os.environ.get("PROBE_NO_MASKS"): sys.exit(0)
splits = {"xcd 0-3 | 4-7 (bit i -> xcd i % 8)": (lambda i: i % 8 < 4, lambda i: i % 8 >= 4),
          "even | odd xcds": (lambda i: i % 2 == 0, lambda i: i % 2 == 1),
          "low | high half of every xcd (bit i -> cu i // 8)": (lambda i: i // 8 < 16, lambda i: i // 8 >= 16),
          "contiguous halves (bits 0-127 | 128-255)": (lambda i: i < 128, lambda i: i >= 128)}
for name, (m0, m1) in splits.items():
    s0, s1 = masked_stream(m0), masked_stream(m1)
    print("%-52s both %.2f ms | leres alone on half %.2f ms, rtmdet+isnet alone on half %.2f ms"
          % (name, wall(two(s0, s1)), wall(one(s1, b, tb)), wall(lambda: (one(s0, a, ta)(), one(s0, i_, ti)()))), flush=True)
