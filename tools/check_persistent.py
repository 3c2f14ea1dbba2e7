#!/usr/bin/env python3
"""every distinct conv layer of a net (RTMDet n=8 by default; argv: rtmdet|leres|isnet [batch]) as a one-op program: the persistent tile
configurations (38..49, serial split-K forced) against configuration 6 with parallel split-K, bit for bit.
REPS=<n> repeats every configuration n times (rare races), ONLY3x3=1 keeps the stride-1 3x3 layers on maps of 40 px and more"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CSM_AUTOTUNE"] = "0"
import numpy as np, torch
from cartoonsegmentation_amd import nets, _lib
from cartoonsegmentation_amd.program import Program
from cartoonsegmentation_amd.runtime import CompiledProgram
from cartoonsegmentation_amd.weights import SynthWeights
which = sys.argv[1] if len(sys.argv) > 1 else 'rtmdet'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
if which == 'rtmdet':
    prog = nets.build_rtmdet(SynthWeights('rtmdet.'), B, 640, 640)[0].prog
elif which == 'leres':
    prog = nets.build_leres(SynthWeights('leres.'), B, 640, 640)
else:
    prog = nets.build_isnet(SynthWeights('isnet.'), 2 * B, 720, 720)
L = _lib.load()
seen = set()
bad = 0
for o in prog.ops:
    if o['kind'] != 1 or (o['flags'] & 2):
        continue
    vi, vo = prog.views[o['in0']], prog.views[o['out']]
    nat = o['nat']
    cin, cout = nat['cin_g'] * nat['groups'], nat['cout_g'] * nat['groups']
    sig = (vi.n, vi.h, vi.w, cin, cout, o['kh'], o['stride'], o['dil'], nat['groups'], o['pad'])
    if sig in seen or cin % 32 or (os.environ.get('ONLY3x3') and not (o['kh'] == 3 and o['stride'] == 1 and vi.h >= 40)):
        continue
    seen.add(sig)
    n, h, w, cin, cout, k, s, d, g, pad = sig
    p = Program("l"); x = p.buffer(n, h, w, cin); x.buf.first = 0
    W = (np.random.default_rng(1).standard_normal((cout, cin // g, k, k)) * 0.05).astype(np.float32)
    y = p.conv(x, W, np.linspace(-1, 1, cout).astype(np.float32), stride=s, pad=pad, dil=d, groups=g, act='relu'); y.buf.keep = True
    p.plan()
    cp = CompiledProgram(p, 'cuda'); cp.workspace.normal_()
    keep = cp.workspace.clone()
    def run(cfg, ser):
        cp.workspace.copy_(keep)
        L.csm_debug_force_conv_cfg(cfg); L.csm_debug_force_splitk_serial(ser); cp.run()
        return cp.read_view(y).clone()
    ref = run(6, 0)
    res = []
    for cfg in range(38, 53):
        out = run(cfg, 1)
        for _ in range(int(os.environ.get('REPS', '1')) - 1):
            o2 = run(cfg, 1)
            if not torch.equal(o2, ref):
                out = o2
        ok = bool(torch.equal(ref, out))
        res.append(ok)
        if not ok:
            bad += 1
            diff = (ref != out)
            idx = diff.nonzero()[0].tolist()
            print("  MISMATCH cfg %d: %d of %d values differ, first at %s ref %g got %g" % (cfg, int(diff.sum()), diff.numel(), idx, float(ref[tuple(idx)]), float(out[tuple(idx)])))
    print("%s ksplit %d: %s" % (sig, p.ops[0]['ksplit'], "".join("." if r else "X" for r in res)), flush=True)
L.csm_debug_force_conv_cfg(-1); L.csm_debug_force_splitk_serial(-1)
print("mismatching (layer, cfg) pairs:", bad)
