"""CPU baseline of one seg+depth+warp frame (TEST INFRASTRUCTURE / bench.py cpu_baseline leg ONLY).

Runs the oracle pipeline (oracle/segment.py, oracle/nets.py, oracle/warp.py) on a REDUCED frame and scales the time
to the 1024x1024 workload by the algorithmic conv FLOP ratio (convolutions are >99% of the CPU time)."""
import os
import time

import numpy as np


def cpu_baseline(seconds_budget=20.0, frame=256, det=192, depth=192, refine=192, instances=2):
    from cartoonsegmentation_amd import synth
    from cartoonsegmentation_amd.nets import build_isnet, build_leres, build_rtmdet
    from cartoonsegmentation_amd.weights import SynthWeights
    from . import nets as onets, segment as oseg, warp as owarp
    img = synth.image_u8(frame, frame, 1234)
    rp, cfg = build_rtmdet(SynthWeights('rtmdet.'), 1, det, det)
    cfg.max_per_img = instances
    isn = build_isnet(SynthWeights('isnet.'), instances, refine, refine)
    ler = build_leres(SynthWeights('leres.'), 1, depth, depth)
    t0 = time.perf_counter()
    d = oseg.detect(img, rp, cfg, det, 0.3)
    n = d.get('n', 0)
    if n:
        oseg.refine(img, d['masks'][:instances], lambda b: isn if b == instances else build_isnet(SynthWeights('isnet.'), b, refine, refine), refine, 0.3)
    x = np.random.default_rng(0).normal(0, 1, (1, 3, depth, depth)).astype(np.float32)
    y = np.zeros((1, 1, depth, depth), np.float32)
    onets.run_program(ler, [x, y])
    sc = synth.warp_scene(frame, frame, 1234)
    _, dep, _, pts, _ = owarp.disparity_to_points(sc['disp'], sc['focal'], sc['baseline'])
    owarp.warp_frame(pts.reshape(1, 3, -1), np.concatenate([sc['rgb'], dep.reshape(1, 1, -1)], 1), frame, frame, sc['focal'],
                     sc['baseline'], np.array([3.0, -2.0, -10.0], np.float32), 1)
    dt = time.perf_counter() - t0
    fl_small = rp.prog.flops + isn.flops + ler.flops
    # conv FLOPs scale with pixel count: scale each net from its reduced size to the benchmark's size
    full = rp.prog.flops * (640.0 / det) ** 2 + isn.flops * (720.0 / refine) ** 2 + ler.flops * (640.0 / depth) ** 2
    est = dt * full / fl_small
    threads = int(os.environ.get('OMP_NUM_THREADS', os.cpu_count() or 1))
    return {"value": round(1.0 / est, 5), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "oracle pipeline on a %dx%d frame (det %d, refine %d x%d inst, LeReS %d): %.1f s for %.1f GFLOP; "
                      "scaled by conv FLOPs to the 1024x1024 workload (%.0f GFLOP)" % (frame, frame, det, refine, instances, depth, dt,
                                                                                      fl_small / 1e9, full / 1e9)}
