#!/usr/bin/env python3
"""how much of the 8-frame step is host-side serialisation?  The same frames as the bench's step, computed by 1 / 2 / 4 host threads with
their own pipelines and streams (FrameLanes): python tools/lanes_probe.py "8x1 4x2 2x4 8x2" """
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    combos = [tuple(int(v) for v in c.split('x')) for c in (sys.argv[1] if len(sys.argv) > 1 else "8x1 4x2 2x4").split()]
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    wl = bench.FrameWorkload(1024, 0, dev, batch=8)
    for _ in range(2):
        wl.step()
    torch.cuda.synchronize()
    print(json.dumps({"serial_batch8": wl._fps(batch=8, steps=4)}), flush=True)
    for b, l in combos:
        print(json.dumps(wl._fps_lanes(b, l, steps=4)), flush=True)


if __name__ == '__main__':
    main()
