"""CPU baseline of one seg+depth+warp frame (TEST INFRASTRUCTURE / bench.py cpu_baseline leg ONLY).

Times the oracle (oracle/nets_oracle.c with OpenMP over output pixels, oracle/warp_oracle.c) on the benchmark's OWN sizes:
RTMDet-Ins-L @640 (batch 1), ISNet @720 (one run per instance), LeReS @640 and one 1024 x 1024 warp frame -- one frame's worth of
every dense stage, each timed by itself (the mask head, mask resize and depth glue are < 1 % of a frame and are not timed: the
baseline is, if anything, flattered).  Nothing is extrapolated unless the time budget
runs out first (then the remaining nets are scaled from the measured GFLOP/s and the sample string says so)."""
import os
import time

import numpy as np


def cpu_baseline(seconds_budget=90.0, frame=1024, det=640, depth=640, refine=720, instances=2):
    from cartoonsegmentation_amd import synth
    from cartoonsegmentation_amd.nets import build_isnet, build_leres, build_rtmdet
    from cartoonsegmentation_amd.weights import SynthWeights
    from . import nets as onets, warp as owarp
    threads = int(os.environ.get('OMP_NUM_THREADS', os.cpu_count() or 1))
    rng = np.random.default_rng(0)
    stages, spent, gflops_rate = [], 0.0, None

    def run_net(name, prog_fn, shapes, reps=1):
        nonlocal spent, gflops_rate
        prog = prog_fn()
        gf = prog.flops / 1e9 * reps
        if gflops_rate is not None and spent + gf / gflops_rate > seconds_budget:
            stages.append((name, gf / gflops_rate, gf, False))
            spent += gf / gflops_rate
            return
        ext = [rng.normal(0, 1, s).astype(np.float32) for s in shapes]
        t0 = time.perf_counter()
        onets.run_program(prog, ext)
        dt = (time.perf_counter() - t0) * reps
        stages.append((name, dt, gf, True))
        spent += dt
        gflops_rate = sum(s[2] for s in stages if s[3]) / max(sum(s[1] for s in stages if s[3]), 1e-9)

    run_net('rtmdet@%d' % det, lambda: build_rtmdet(SynthWeights('rtmdet.'), 1, det, det)[0].prog, [(1, 3, det, det)])
    run_net('isnet@%d x%d' % (refine, instances), lambda: build_isnet(SynthWeights('isnet.'), 1, refine, refine),
            [(1, 4, refine, refine), (1, 1, refine, refine)], reps=instances)
    run_net('leres@%d' % depth, lambda: build_leres(SynthWeights('leres.'), 1, depth, depth), [(1, 3, depth, depth), (1, 1, depth, depth)])
    sc = synth.warp_scene(frame, frame, 1234)
    t0 = time.perf_counter()
    _, dep, _, pts, _ = owarp.disparity_to_points(sc['disp'], sc['focal'], sc['baseline'])
    owarp.warp_frame(pts.reshape(1, 3, -1), np.concatenate([sc['rgb'], dep.reshape(1, 1, -1)], 1), frame, frame, sc['focal'],
                     sc['baseline'], np.array([3.0, -2.0, -10.0], np.float32), 1)
    t_warp = time.perf_counter() - t0
    total = sum(s[1] for s in stages) + t_warp
    parts = ", ".join("%s %.2f s%s" % (n, t, "" if m else " (scaled from %.0f GFLOP/s)" % gflops_rate) for n, t, g, m in stages)
    return {"value": round(1.0 / total, 5), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "one %dx%d frame at the benchmark's sizes, each stage timed once on the host (oracle, OpenMP, %d threads): %s, "
                      "warp %dx%d %.2f s (1 thread); %.0f GFLOP of convolutions in total"
                      % (frame, frame, threads, parts, frame, frame, t_warp, sum(s[2] for s in stages))}
