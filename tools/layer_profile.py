#!/usr/bin/env python3
"""per-op profile of a lowered net on the GPU (HIP events per op): prints the slowest layers and TF/s per conv"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cartoonsegmentation_amd import nets, _lib
if os.environ.get('LP_LIB'):                    # A/B of two builds of the library (e.g. make PREFETCH=1 OUT=...)
    _lib.LIB_PATH = os.path.abspath(os.environ['LP_LIB'])
if os.environ.get('LP_TUNER_OPTIONS'):
    _lib.load().csm_debug_conv_tuner_options(int(os.environ['LP_TUNER_OPTIONS']))
from cartoonsegmentation_amd.runtime import CompiledProgram
from cartoonsegmentation_amd.weights import SynthWeights

KIND = {1: 'conv', 2: 'dwconv', 3: 'maxpool', 4: 'bilinear', 5: 'nearest', 6: 'add', 7: 'gavg', 8: 'scale', 9: 'to_nhwc', 10: 'to_nchw', 11: 'act', 12: 'copy'}


TOP = int(os.environ.get('LP_TOP', 14))
B = int(os.environ.get('LP_BATCH', 1))


def prof(name, prog, ext, top=TOP):
    cp = CompiledProgram(prog, 'cuda')
    cp.run(*ext)
    ms = None
    for _ in range(3):
        m = cp.profile(*ext)
        ms = m if ms is None else [min(a, b) for a, b in zip(ms, m)]
    by_kind = {}
    rows = []
    for t, o in zip(ms, prog.ops):
        k = KIND[o['kind']]
        by_kind[k] = by_kind.get(k, 0.0) + t
        v_in, v_out = prog.views[o['in0']], prog.views[o['out']]
        fl = 0
        if o['kind'] == 1:
            nat = o['nat']
            fl = 2 * v_out.n * v_out.h * v_out.w * nat['cout_g'] * nat['groups'] * nat['cin_g'] * o['kh'] * o['kw']
        rows.append((t, k, "%dx%dx%dx%d->%dx%dx%d k%d s%d d%d g%d S%d T%d" % (v_in.n, v_in.h, v_in.w, v_in.c, v_out.h, v_out.w, v_out.c, o['kh'], o['stride'], o['dil'], o['groups'], o.get('ksplit', 1), cp.ops[len(rows)].tile - 1) + (" WINO4" if o['kind'] == 1 and o['flags'] & 8 else (" WINO" if o['kind'] == 1 and o['flags'] & 4 else "")), fl))
    tot = sum(ms)
    print("== %s: %.3f ms total (event-bracketed), %.1f GFLOP -> %.1f TF/s ; by kind: %s" % (name, tot, prog.flops / 1e9, prog.flops / tot / 1e9,
          ", ".join("%s %.2f" % kv for kv in sorted(by_kind.items(), key=lambda kv: -kv[1]))))
    agg = {}
    for t, k, d, fl in rows:
        a = agg.setdefault((k, d), [0.0, 0, 0]); a[0] += t; a[1] += 1; a[2] += fl
    for (k, d), (t, n, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        print("   %7.3f ms  x%-3d %-7s %-52s %6.1f TF/s" % (t, n, k, d, fl / t / 1e9 if t > 0 else 0))
    if os.environ.get('LP_NONCONV'):            # the memory-bound ops: bytes moved (input view + output view) and TB/s
        agg = {}
        for t, o in zip(ms, prog.ops):
            if o['kind'] == 1:
                continue
            vi, vo = prog.views[o['in0']], prog.views[o['out']]
            by = 4.0 * (vi.n * vi.h * vi.w * vi.c + vo.n * vo.h * vo.w * vo.c) + (4.0 * vo.n * vo.h * vo.w * vo.c if o['kind'] == 6 else 0.0)
            a = agg.setdefault((KIND[o['kind']], "%dx%dx%dx%d->%dx%dx%d k%d s%d" % (vi.n, vi.h, vi.w, vi.c, vo.h, vo.w, vo.c, o['kh'], o['stride'])), [0.0, 0, 0.0])
            a[0] += t; a[1] += 1; a[2] += by
        for (k, d), (t, n, by) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
            print("   %7.3f ms  x%-3d %-8s %-44s %6.1f MB each  %5.2f TB/s" % (t, n, k, d, by / n / 1e6, by / t / 1e9))


if __name__ == '__main__':
    which = sys.argv[1:] or ['isnet', 'leres', 'rtmdet']
    dev = 'cuda'
    if 'isnet' in which:
        nb = min(2 * B, 16)
        p = nets.build_isnet(SynthWeights('isnet.'), nb, 720, 720)
        prof('isnet n=%d 720' % nb, p, [torch.rand(nb, 4, 720, 720, device=dev), torch.empty(nb, 1, 720, 720, device=dev)])
    if 'leres' in which:
        p = nets.build_leres(SynthWeights('leres.'), B, 640, 640)
        prof('leres n=%d 640' % B, p, [torch.randn(B, 3, 640, 640, device=dev), torch.empty(B, 1, 640, 640, device=dev)])
    if 'inpaint' in which:                      # the GridNet of the point-cloud inpainting at 1024^2 (one of the two passes of a video)
        from cartoonsegmentation_amd.nets.inpaint import build_inpaint_context, build_inpaint_grid
        ws = SynthWeights('inpaint.')
        p = build_inpaint_context(ws, 1024, 1024)
        prof('inpaint context 1024', p, [torch.randn(1, 4, 1024, 1024, device=dev), torch.empty(1, 64, 1024, 1024, device=dev)])
        p = build_inpaint_grid(ws, 1024, 1024)
        prof('inpaint grid 1024', p, [torch.randn(1, 69, 1024, 1024, device=dev), torch.empty(1, 3, 1024, 1024, device=dev), torch.empty(1, 1, 1024, 1024, device=dev)])
    if 'rtmdet' in which:
        rp, _ = nets.build_rtmdet(SynthWeights('rtmdet.'), B, 640, 640)
        prof('rtmdet n=%d 640' % B, rp.prog, [torch.randn(B, 3, 640, 640, device=dev)])
