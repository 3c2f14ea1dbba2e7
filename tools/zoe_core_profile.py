#!/usr/bin/env python3
"""per-kind profile of the built-in DPT-BEiT-L core of ZoeDepth (HIP events per op): argv = height width [batch] (default 672 672 1);
ATT_OPTIONS=<bits> sets csm_debug_attention_options (A/B of the attention kernel's variants)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cartoonsegmentation_amd.nets import build_dpt_beit
from cartoonsegmentation_amd.runtime import CompiledProgram
from cartoonsegmentation_amd.weights import SynthWeights
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (672, 672)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
KIND = {1: 'conv', 4: 'bilinear', 6: 'add', 9: 'to_nhwc', 10: 'to_nchw', 11: 'act', 15: 'layernorm', 16: 'attention', 17: 'tokens', 18: 'depth_to_space'}
p = build_dpt_beit(SynthWeights('zoe.core.core.'), B, H, W)
cp = CompiledProgram(p, 'cuda')
if os.environ.get('ATT_OPTIONS'):
    from cartoonsegmentation_amd import _lib
    _lib.load().csm_debug_attention_options(int(os.environ['ATT_OPTIONS']))
ext = [torch.randn(b.n, b.c, b.h, b.w, device='cuda') for b in sorted((b for b in p.bufs if b.ext >= 0), key=lambda b: b.ext)]
cp.run(*ext)
ms = None
for _ in range(3):
    m = cp.profile(*ext)
    ms = m if ms is None else [min(a, b) for a, b in zip(ms, m)]
by = {}
for t, o in zip(ms, p.ops):
    k = KIND.get(o['kind'], str(o['kind']))
    by[k] = by.get(k, 0.0) + t
N = (H // 16) * (W // 16) + 1
att_fl = sum(4 * B * o['groups'] * N * N * o['cin_g'] for o in p.ops if o['kind'] == 16)
conv_fl = p.flops - att_fl
print("dpt-beit-l %dx%d n=%d (%d tokens): %.3f ms, %.1f GFLOP -> %.1f TF/s; by kind: %s" % (H, W, B, N, sum(ms), p.flops / 1e9, p.flops / sum(ms) / 1e9,
      ", ".join("%s %.3f" % kv for kv in sorted(by.items(), key=lambda kv: -kv[1]))))
print("  conv %.1f TF/s, attention %.1f TF/s (%.1f us per layer)" % (conv_fl / by['conv'] / 1e9, att_fl / by['attention'] / 1e9, by['attention'] * 1e3 / 24))
