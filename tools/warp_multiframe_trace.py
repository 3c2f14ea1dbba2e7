#!/usr/bin/env python3
"""Independent check of the multi-frame warp figure (bench.py: warp_chain.tiled_1024_multiframe) from a rocprofv3 kernel trace.

  probe (run under `rocprofv3 --kernel-trace --output-format csv -d DIR -o mf -- python tools/warp_multiframe_trace.py probe`):
      REPS calls of csm_warp_frames_tiled with K frames of one 1024^2 cloud, a synchronize + 50 ms of sleep between calls, and the HIP-event
      time of every call printed (the caller's stream: what bench.py reports);
  span (python tools/warp_multiframe_trace.py span DIR):
      reads the kernel trace, splits it into calls at gaps > 10 ms, and prints for every call with 3 K tile kernels
      (first kernel start -> last kernel end) / K -- the frame rate as the GPU's own timestamps see it -- next to the sum of the kernel
      durations / K (what one stream would take without the overlap)."""
import csv
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = int(os.environ.get('MF_K', 60))
REPS = int(os.environ.get('MF_REPS', 6))
SIZE = int(os.environ.get('MF_SIZE', 1024))


def probe():
    import torch
    from cartoonsegmentation_amd import ops, synth
    dev = 'cuda'
    sc = synth.warp_scene(SIZE, SIZE, 1234)
    disp = torch.from_numpy(sc['disp']).to(dev)
    disp = disp / disp.max() * sc['baseline']
    depth, _, pts, _ = ops.disparity_to_points(disp, sc['focal'], sc['baseline'])
    pts, dep, rgb = pts.view(1, 3, -1).contiguous(), depth.view(1, 1, -1).contiguous(), torch.from_numpy(sc['rgb']).to(dev)
    loc = int(depth.argmin().item())
    settings, common = synth.shift_request(sc, float(depth.min().item()), (loc % SIZE, loc // SIZE))
    shift = ops.shift_vector(settings, common)
    wf = ops.WarpFrame(SIZE, SIZE, dev, path='tiled')
    out = torch.empty((K, SIZE, SIZE, 3), dtype=torch.uint8, device=dev)
    shifts = [shift] * K
    for rep in range(REPS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        time.sleep(0.05)
        e0.record()
        wf.frames(pts, rgb, dep, sc['focal'], sc['baseline'], shifts, lanes=3, out=out)
        e1.record()
        e1.synchronize()
        print("call %d: %d frames, HIP events on the caller's stream: %.2f us per frame" % (rep, K, e0.elapsed_time(e1) * 1e3 / K), flush=True)


def span(d):
    rows = []
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    calls, cur = [], []
    for r in rows:
        if cur and r[0] - max(e for _, e, _ in cur) > 10_000_000:
            calls.append(cur)
            cur = []
        cur.append(r)
    if cur:
        calls.append(cur)
    n = 0
    for c in calls:
        tiles = [r for r in c if 'k_tile_' in r[2]]
        if len(tiles) < 3 * K:
            continue
        t0, t1 = min(s for s, _, _ in tiles), max(e for _, e, _ in tiles)
        busy = sum(e - s for s, e, _ in tiles)
        by = {}
        for s, e, name in tiles:
            k = name[name.find('k_tile_'):].split('(')[0].split('<')[0]
            by.setdefault(k, [0, 0])
            by[k][0] += 1
            by[k][1] += e - s
        print("call %d: %d tile kernels, first start -> last end %.1f us = %.2f us per frame (%d frames); sum of kernel durations %.2f us per frame; %s"
              % (n, len(tiles), (t1 - t0) / 1e3, (t1 - t0) / 1e3 / K, K, busy / 1e3 / K,
                 ", ".join("%s x%d avg %.1f us" % (k, v[0], v[1] / v[0] / 1e3) for k, v in sorted(by.items()))))
        n += 1


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == 'span':
        span(sys.argv[2])
    else:
        probe()
