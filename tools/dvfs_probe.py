#!/usr/bin/env python3
"""does the conv engine run at the nominal clock?  Loops one layer for a few seconds (real data, then all-zero operands: same
instruction stream, minimum switching power) while a thread samples shader clock and socket power through rocm-smi.
args: n h w cin cout k [seconds]"""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cartoonsegmentation_amd.program import Program
from cartoonsegmentation_amd.runtime import CompiledProgram

v = [int(x) for x in sys.argv[1:]]
n, h, w, cin, cout, k = v[:6]
secs = v[6] if len(v) > 6 else 4
p = Program("l"); x = p.buffer(n, h, w, cin)
W = (np.random.default_rng(0).standard_normal((cout, cin, k, k)) * 0.05).astype(np.float32)
p.conv(x, W, np.zeros(cout, np.float32), stride=1, pad=k // 2, act='relu'); p.plan()
cp = CompiledProgram(p, 'cuda'); cp.workspace.normal_(); cp.run()


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(r); c = d[next(iter(d))]
            sclk = [v for kk, v in c.items() if 'sclk' in kk.lower()]
            pw = [v for kk, v in c.items() if 'power' in kk.lower()]
            out.append((sclk[0] if sclk else '?', pw[0] if pw else '?'))
        except Exception as e:
            out.append(('err', str(e)[:40]))


for label in ('normal', 'zero'):
    if label == 'zero':
        cp.workspace.zero_(); cp.weights.zero_()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time(); reps = 0
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(50):
            cp.run()
        reps += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    ms = e0.elapsed_time(e1) / reps
    print("%-6s %.1f us per layer, %.1f TF/s; samples (sclk, W): %s" % (label, ms * 1e3, p.flops / ms / 1e9, out[1:-1][:8]), flush=True)
