#!/usr/bin/env python3
"""bench.py -- frames/s of the MI355X hot path (BASELINE.json metric) on N GPUs of one node.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" is one pass of the hot path over `--batch` (default 8) 1024x1024 frames per rank (weak
scaling: every rank owns its own frames, 8 per rank = BASELINE configs[3]'s 64 frames on 8 GPUs;
SURVEY.md 8e; outputs are gathered to rank 0 over RCCL inside the timed region).  Inputs are synthetic, seeded and already resident in HBM when timing starts.
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant kernel,
HIP-event timed on the launch stream) and `cpu_baseline` (the oracle timed on this host).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3  # same guide: fp32-input MFMA dense peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)      # 1.4 s timed at 8 frames per step: the 0.7 s of 8 steps moved +-1.5 % with the box's clock state
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--workload", default="auto", help="auto | warp | frame")
    ap.add_argument("--batch", type=int, default=8, help="frames per GPU per step (frame workload); 8 per rank = BASELINE configs[3] "
                                                        "(64 frames on 8 GPUs), and the same per-rank work at every N (weak scaling)")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra single-GPU measurements (batch / instances / det size / video)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-iou", action="store_true", help="skip the one-frame mask IoU against the CPU oracle (about 20 s of host time, untimed)")
    ap.add_argument("--no-roofline", action="store_true", help="profiling runs: skip the per-op HIP-event pass, so that the process executes "
                                                               "exactly 1 + warmup + steps steps (rocprofv3 totals divide cleanly)")
    ap.add_argument("--cpu-seconds", type=float, default=90.0, help="budget of the host-side oracle run (full-size nets; stages that "
                                                                    "would not fit are scaled from the measured GFLOP/s and say so)")
    return ap.parse_args()


def event_time_ms(fn, iters, warm=3):
    """average duration of fn() measured with HIP events on torch's current stream (= launch stream)"""
    for _ in range(warm):
        fn()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    en.synchronize()
    return st.elapsed_time(en) / iters


class Workload:
    frames_per_step = 1
    dtype = "f32"
    MAX_INST = 0                 # instance masks shipped per frame (bit-packed); 0 = frames only

    def make_records(self, device, nmax):
        """per-frame output records (cartoonsegmentation_amd/shard.py: uint8 frame | MAX_INST bit-packed masks | count): the frame
        loop writes its uint8 output straight into the record, so shipping a step is ONE gather of one flat tensor per rank"""
        from cartoonsegmentation_amd import shard
        self.fb, self.mb, self.rb = shard.record_layout(self.H, self.W, self.MAX_INST)
        self.records = torch.zeros((nmax, self.rb), dtype=torch.uint8, device=device)
        # device-resident count words 0 .. MAX_INST + 1 (a frame's count is copied device-to-device: no pageable H2D in the loop)
        self.count_words = torch.arange(0, 4096, dtype=torch.int64).view(-1, 1).view(torch.uint8).to(device)

    def frame_slot(self, k):
        return self.records[k, :self.fb].view(self.H, self.W, 3)

    def attach(self, world, dist, device):
        self.world, self.dist = world, dist
        self.gather_list = None
        if dist is not None and dist.get_rank() == 0:
            self.gather_list = [torch.empty((self.frames_per_step, self.rb), dtype=torch.uint8, device=device) for _ in range(world)]

    def step_and_gather(self):
        """one step on this rank, then the per-rank gather of the output records (uint8 frames + bit-packed instance masks, SURVEY.md
        8e) to rank 0.  The gather runs asynchronously on the communicator's stream from one of two staging copies, so step s + 1
        computes while the records of step s travel; finish_gathers() (inside the timed region) waits for whatever is still in flight."""
        frame = self.step()
        if self.dist is not None:
            if not hasattr(self, '_stage'):
                self._stage = [torch.empty((self.frames_per_step, self.rb), dtype=torch.uint8, device=frame.device) for _ in range(2)]
                self._work, self._parity = [None, None], 0
            p = self._parity
            self._parity ^= 1
            if self._work[p] is not None:
                self._work[p].wait()                           # the previous gather from this staging buffer (two steps ago)
            self._stage[p].copy_(self.records[:self.frames_per_step])
            self._work[p] = self.dist.gather(self._stage[p], self.gather_list, dst=0, async_op=True)
        return frame

    def finish_gathers(self):
        for p, w in enumerate(getattr(self, '_work', [])):
            if w is not None:
                w.wait()
                self._work[p] = None

    def check_gathered(self):
        """rank 0, after the timed region: the last step's records of every rank unpack into a frame and the shipped masks"""
        from cartoonsegmentation_amd import shard
        if self.gather_list is None:
            return None
        ok, n_masks = True, 0
        for r, g in enumerate(self.gather_list):
            frame, masks, n = shard.read_record(g[0], self.H, self.W, self.MAX_INST)
            ok = ok and frame.shape == (self.H, self.W, 3) and 0 <= n <= 4095 and masks.shape[0] == min(n, self.MAX_INST)
            ok = ok and (masks.shape[0] == 0 or bool(masks.flatten(1).any(1).all()))
            n_masks += int(masks.shape[0])
        return {"records_ok": bool(ok), "masks_in_first_frames": n_masks, "record_bytes": self.rb,
                "gathered_bytes_per_rank_step": self.rb * self.frames_per_step}

    def extra(self):
        return {}


class WarpWorkload(Workload):
    """process_shift -> render_pointcloud(C=4) -> fill_disocclusion -> uint8 (kenburns_effect.py:1027-1040)"""
    name = "kenburns-warp-frame"

    def __init__(self, size, rank, device):
        from cartoonsegmentation_amd import ops, synth
        self.ops, self.H, self.W = ops, size, size
        sc = synth.warp_scene(size, size, 1234 + rank)
        self.scene = sc
        disp = torch.from_numpy(sc['disp']).to(device)
        disp = disp / disp.max() * sc['baseline']                       # kenburns_effect.py:928
        self.depth, self.valid, pts, _ = ops.disparity_to_points(disp, sc['focal'], sc['baseline'])
        self.pts = pts.view(1, 3, -1).contiguous()
        self.rgb = torch.from_numpy(sc['rgb']).to(device)
        self.dep = self.depth.view(1, 1, -1).contiguous()
        dmin = float(self.depth.min().item())
        loc = int(self.depth.argmin().item())
        settings, common = synth.shift_request(sc, dmin, (loc % size, loc // size))
        self.shift = ops.shift_vector(settings, common)
        self.wf = ops.WarpFrame(size, size, device)
        self.P = size * size
        self.N = self.pts.shape[2]
        self.make_records(device, 1)

    def step(self):
        frame, _ = self.wf(self.pts, self.rgb, self.dep, self.scene['focal'], self.scene['baseline'], self.shift)
        if self.dist is not None:
            self.frame_slot(0).copy_(frame)
        return frame

    def config(self, world):
        return {"workload": "warp-only: process_shift+render_pointcloud(C=4,N=P)+fill_disocclusion+uint8, %dx%d"
                            % (self.W, self.H), "frames_per_gpu_step": 1, "parallelism": "frames sharded x%d" % world,
                "note": "PARTIAL frame: seg+depth stages not included in this workload"}

    # B_warp = 155*P bytes per frame for N=P, C=4; per-kernel shares in DESIGN.md
    def algorithmic_bytes(self):
        return 155.0 * self.P

    def extra(self):
        ms = event_time_ms(self.step, 50)
        ach = self.algorithmic_bytes() / (ms * 1e-3) / 1e9
        return {"warp_chain": {"algorithmic_bytes_per_frame": self.algorithmic_bytes(), "us_per_frame": round(ms * 1e3, 2),
                               "achieved_GBps": round(ach, 1), "frac_of_hbm_peak": round(ach / HBM_PEAK_GBS, 4)}}

    def roofline(self):
        """the whole frame chain of csm_warp_frame_tiled (bin, render, holes) against the HBM roof with SURVEY 8(d)'s
        algorithmic bytes (155 P for N = P, C = 4); `traffic` = PMC bytes summed over the chain's kernels (profiles/traffic.json)"""
        ms = event_time_ms(self.step, 100, warm=10)
        alg = self.algorithmic_bytes()
        ach = alg / (ms * 1e-3) / 1e9
        tiled = getattr(self.wf, 'path', 'tiled') == 'tiled'
        return {"bound": "hbm", "kernel": "csm_warp_frame_tiled: k_tile_bin + k_tile_render + k_tile_holes" if tiled
                else "csm_warp_frame: k_fill + k_update_zee + k_degrid + k_update_output + k_finalize_frame + k_fill_holes",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": load_traffic("warp_chain_tiled" if tiled else "k_update_output"), "traffic_source": TRAFFIC_SOURCE,
                "algorithmic_bytes_per_launch": alg,
                "launch_us": round(ms * 1e3, 2)}

    def cpu_baseline(self, seconds):
        from oracle import warp as orc
        pts, rgb, dep = self.pts.cpu().numpy(), self.rgb.cpu().numpy(), self.dep.cpu().numpy()
        rgbd = np.concatenate([rgb, dep], 1)
        n, t0 = 0, time.perf_counter()
        while True:
            orc.warp_frame(pts, rgbd, self.H, self.W, self.scene['focal'], self.scene['baseline'],
                           np.asarray(self.shift, np.float32), degrid_mode=1)
            n += 1
            dt = time.perf_counter() - t0
            if dt >= seconds or n >= 64:
                break
        return {"value": round(n / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                "sample": "%d warp frames %dx%d, oracle/warp_oracle.c single thread" % (n, self.W, self.H)}


class FrameWorkload(Workload):
    """BASELINE.json configs[2] with the shipped yaml's nets: one 1024x1024 frame =
    AnimeInsSeg.infer (RTMDet-Ins-L @det 640 + mask head + ISNet refine @720 on `instances` masks)
    + LeReS depth @640 + depth adjustment + disparity->points + ONE Ken Burns warp (+ crop/resize to uint8)."""
    name = "seg+depth+warp"
    INSTANCES = 2
    MAX_INST = 2                  # masks shipped per frame = the instance cap of the synthetic detector

    def __init__(self, size, rank, device, batch=8):
        self.frames_per_step = batch
        os.environ["CSM_SYNTHETIC_WEIGHTS"] = "1"          # no checkpoints exist offline: closed-form weights
        # every rank builds the same closed-form weights (a rank with placeholder zeros finds no instance, builds no ISNet and
        # would take part in fewer broadcasts); the RCCL broadcast from rank 0 then overwrites them, as it would real checkpoints
        from cartoonsegmentation_amd import ops, synth
        self.ops, self.H, self.W, self.device = ops, size, size, device
        self.pipe = self.make_pipe()
        self.all_imgs = [torch.from_numpy(synth.image_u8(size, size, 1234 + 64 * rank + k)).to(device) for k in range(max(batch, 16))]
        self.imgs = self.all_imgs[:batch]
        self.wf = ops.WarpFrame(size, size, device)
        self.make_records(device, max(batch, 16))
        self.n_inst = None

    def step(self):
        self.n_inst = self.run_frames(self.pipe, self.wf, self.imgs, self.records)
        return self.records[:self.frames_per_step, :self.fb]

    def run_frames(self, pipe, wf, imgs, records):
        """the hot path over `imgs` (one step): seg + depth -> point cloud -> one warp + crop/resize per frame, outputs (uint8 frame,
        bit-packed instance masks, count) written into `records`; returns the instance count of the first frame"""
        from cartoonsegmentation_amd._lib import load, ptr, stream_ptr, i32, i64, f32, check
        if len(imgs) == 1:
            kcs = [pipe.generate_kenburns_config(imgs[0])]     # seg (main stream) || LeReS (side stream)
        else:
            kcs = pipe.generate_kenburns_configs(imgs)         # batched detector / refine / LeReS, per-frame glue
        for k, kc in enumerate(kcs):
            W, H = kc['intWidth'], kc['intHeight']
            d_from = kc['objDepthrange'][0]
            shift = self.ops.shift_vector({'fltShiftU': 30.0, 'fltShiftV': -20.0, 'fltDepthFrom': d_from, 'fltDepthTo': d_from / 1.25}, kc)
            frame, _ = wf(kc['tenInpaPoints'], kc.inpainted_img, kc['tenInpaDepth'], kc['fltFocal'], kc['fltBaseline'], shift)
            pw, ph = int(0.97 * W), int(0.97 * H)
            rec = records[k]
            check(load().csm_crop_resize_u8(ptr(frame), i32(H), i32(W), i32(ph), i32(pw), f32(W / 2.0), f32(H / 2.0), ptr(rec), stream_ptr()))
            # the frame's instance masks join its output record bit-packed (frame + packed masks travel to rank 0)
            masks = kc.instances.masks
            n = 0 if kc.instances.is_empty else int(masks.shape[0])
            for j in range(min(n, self.MAX_INST)):
                check(load().csm_pack_mask_bits(ptr(masks[j].view(torch.uint8)), i64(H * W), ptr(rec[self.fb + j * self.mb:]), stream_ptr()))
            if n < self.MAX_INST:
                rec[self.fb + n * self.mb: self.fb + self.MAX_INST * self.mb].zero_()
            rec[self.fb + self.MAX_INST * self.mb:].copy_(self.count_words[min(n, 4095)])
        return len(kcs[0].instances)

    def make_pipe(self):
        from cartoonsegmentation_amd.kenburns import KenBurnsConfig, KenBurnsPipeline
        size = self.H
        cfg = KenBurnsConfig(det_ckpt='synthetic', depth_est='leres', depth_est_size=640, max_size=size,
                             refine_crf=False, depth_field=False, focal=size / 2.0,
                             mask_refine_kwargs={'refine_method': 'refinenet_isnet', 'refine_size': 720})
        pipe = KenBurnsPipeline(cfg, device=str(self.device))
        pipe.max_instances = self.INSTANCES          # synthetic weights score every prior ~0.49: cap like infer(max_instances=)
        pipe.overlap_depth = os.environ.get("CSM_OVERLAP_DEPTH", "1") == "1"
        return pipe

    def _fps_lanes(self, batch, lanes, steps=4):
        """frames/s with `lanes` steps in flight (cartoonsegmentation_amd/lanes.py): every lane is a thread with its own pipeline
        object, streams, warp scratch and output records; a step's frames are computed exactly as in step()"""
        from cartoonsegmentation_amd.lanes import FrameLanes
        imgs_all = self.all_imgs

        def make_worker(i):
            pipe, wf = self.make_pipe(), self.ops.WarpFrame(self.H, self.W, self.device)
            records = torch.zeros((batch, self.rb), dtype=torch.uint8, device=self.device)
            self.run_frames(pipe, wf, imgs_all[:batch], records)          # builds this lane's programs (tiles come from the shared table)
            torch.cuda.current_stream().synchronize()
            return lambda imgs: self.run_frames(pipe, wf, imgs, records)
        self._fps(batch=batch, steps=1)                                   # the main pipeline has tuned every layer of this batch size
        fl = FrameLanes(make_worker, lanes, self.device)
        try:
            jobs = [[imgs_all[(j * batch + q) % len(imgs_all)] for q in range(batch)] for j in range(lanes * steps)]
            fl.map(jobs[:lanes]); torch.cuda.synchronize()
            t0 = time.perf_counter()
            fl.map(jobs)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        finally:
            fl.close()
        return {"frames_per_s": round(len(jobs) * batch / dt, 2), "ms_per_frame": round(dt / (len(jobs) * batch) * 1e3, 3), "batch": batch,
                "lanes": lanes, "what": "%d steps of %d frame(s) in flight on %d host threads / stream sets (FrameLanes); each frame "
                                        "computed as in the serial loop" % (lanes, batch, lanes)}

    def _fps_host_fed(self, steps=6):
        """the headline step with the host feed a real multi-rank run has: every step's input frames start in PINNED HOST
        memory and are uploaded on a copy stream (two device buffers: the upload of step s + 1 runs under the compute of step s), and
        the step's output records (uint8 frames + bit-packed masks) are copied back to pinned host memory the same way.  `value` of the
        bench line keeps the inputs resident (the contract); this is the PCIe-inclusive rate beside it."""
        B, dev = self.frames_per_step, self.device
        host_in = [torch.stack([im.cpu() for im in self.all_imgs[(p * B) % len(self.all_imgs):][:B]]).pin_memory() for p in range(2)]
        dev_in = [torch.empty_like(h, device=dev) for h in host_in]
        host_out = [torch.empty((B, self.rb), dtype=torch.uint8).pin_memory() for _ in range(2)]
        copy, main = torch.cuda.Stream(dev), torch.cuda.current_stream(dev)
        up = [torch.cuda.Event() for _ in range(2)]
        done = [torch.cuda.Event() for _ in range(2)]
        recs = [torch.zeros((B, self.rb), dtype=torch.uint8, device=dev) for _ in range(2)]

        def upload(p):
            with torch.cuda.stream(copy):
                copy.wait_event(done[p])                           # the step that read this buffer two steps ago has finished
                dev_in[p].copy_(host_in[p], non_blocking=True)
                up[p].record(copy)

        def run(n):
            upload(0)
            for s_ in range(n):
                p = s_ & 1
                if s_ + 1 < n:
                    upload(p ^ 1)
                main.wait_event(up[p])
                self.run_frames(self.pipe, self.wf, list(dev_in[p]), recs[p])
                done[p].record(main)
                with torch.cuda.stream(copy):
                    copy.wait_event(done[p])
                    host_out[p].copy_(recs[p], non_blocking=True)
            copy.synchronize(); main.synchronize()
        for e in done:
            e.record(main)
        run(2); torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"frames_per_s": round(steps * B / dt, 2), "ms_per_step": round(dt / steps * 1e3, 3), "batch": B,
                "h2d_bytes_per_step": int(host_in[0].numel()), "d2h_bytes_per_step": int(host_out[0].numel()),
                "what": "headline step fed from pinned host memory (H2D of the %d input frames and D2H of their output records on a copy "
                        "stream, double-buffered, inside the timed region)" % B}

    def _zoe_variant(self, frames=3, batch=8):
        """BASELINE configs[2] with its literal depth network: seg + ZoeDepth (built-in MiDaS DPT-BEiT-L core, 672 x 672, flip TTA: the plain
        and the mirrored pass, 1765 tokens each, as the two samples of one core run) + one warp per 1024 x 1024 frame, serial; and the core program's own conv population against the MFMA roof"""
        from cartoonsegmentation_amd.kenburns import KenBurnsConfig, KenBurnsPipeline
        size = self.H
        cfg = KenBurnsConfig(det_ckpt='synthetic', depth_est='zoe', max_size=size, refine_crf=False, depth_field=False, focal=size / 2.0,
                             mask_refine_kwargs={'refine_method': 'refinenet_isnet', 'refine_size': 720})
        t_build = time.perf_counter()
        pipe = KenBurnsPipeline(cfg, device=str(self.device))
        pipe.max_instances = self.INSTANCES
        records = torch.zeros((1, self.rb), dtype=torch.uint8, device=self.device)
        self.run_frames(pipe, self.wf, self.all_imgs[:1], records); torch.cuda.synchronize()       # builds + tunes the core and head programs
        t_build = time.perf_counter() - t_build
        t0 = time.perf_counter()
        for k in range(frames):
            self.run_frames(pipe, self.wf, [self.all_imgs[(k + 1) % len(self.all_imgs)]], records)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / frames
        res = {"frames_per_s": round(1.0 / dt, 2), "ms_per_frame": round(dt * 1e3, 2), "batch": 1, "build_and_tune_s": round(t_build, 1),
               "what": "seg (RTMDet + ISNet) + ZoeDepth on the built-in DPT-BEiT-L core (img_size 672, pad + flip TTA) + 1 warp, 1024x1024, serial"}
        if batch > 1:
            # the same workload as an 8-frame step like the headline (configs[2]-literal network, comparable with `value`): one core run over
            # the 8 frames and their mirrored passes (n = 16 samples of 1765 tokens)
            recs = torch.zeros((batch, self.rb), dtype=torch.uint8, device=self.device)
            imgs = [self.all_imgs[k % len(self.all_imgs)] for k in range(batch)]
            self.run_frames(pipe, self.wf, imgs, recs); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                self.run_frames(pipe, self.wf, imgs, recs)
            torch.cuda.synchronize()
            dtb = (time.perf_counter() - t0) / 2
            res["batch%d" % batch] = {"frames_per_s": round(batch / dtb, 2), "ms_per_step": round(dtb * 1e3, 2), "batch": batch,
                                      "what": "%d frames per step: seg + ZoeDepth (one DPT-BEiT-L core run, n = %d) + %d warps" % (batch, 2 * batch, batch)}
        core = getattr(pipe.depth_zoe, 'core', None)
        for (n, h, w), cp in getattr(core, '_progs', {}).items():
            ext = [torch.randn(b.n, b.c, b.h, b.w, device=self.device) for b in sorted((b for b in cp.prog.bufs if b.ext >= 0), key=lambda b: b.ext)]
            cp.run(*ext)
            ms = None
            for _ in range(3):
                m = cp.profile(*ext)
                ms = m if ms is None else [min(a, b) for a, b in zip(ms, m)]
            conv_ms = sum(t for t, o in zip(ms, cp.prog.ops) if o['kind'] == 1)
            att_ms = sum(t for t, o in zip(ms, cp.prog.ops) if o['kind'] == 16)
            conv_fl = sum(conv_op_flops(cp.prog, o)[0] for o in cp.prog.ops if o['kind'] == 1)
            conv_ex = sum(conv_op_flops(cp.prog, o)[1] for o in cp.prog.ops if o['kind'] == 1)       # Winograd layers execute 16 / 36 (F(2x2)) or 9 / 36 (F(4x4)) of their natural FLOPs
            res["core_%dx%d_n%d" % (h, w, n)] = {"all_ops_ms": round(sum(ms), 3), "conv_ms": round(conv_ms, 3), "attention_ms": round(att_ms, 3),
                                                 "gflop": round(cp.prog.flops / 1e9, 1), "conv_tflops": round(conv_ex / conv_ms / 1e9, 1),
                                                 "conv_frac_of_mfma_peak": round(conv_ex / conv_ms / 1e9 / MFMA_F32_PEAK_TF, 4),
                                                 "conv_tflops_direct_equivalent": round(conv_fl / conv_ms / 1e9, 1),
                                                 "tokens": (h // 16) * (w // 16) + 1}
        del pipe
        torch.cuda.empty_cache()
        return res

    def mask_iou_vs_oracle(self):
        """BASELINE.json's `mask IoU vs ref`: ONE 1024 x 1024 frame through AnimeInsSeg.infer on the GPU and through the CPU oracle pipeline
        (oracle/segment.py, the checker; outside every timed region) -- per-instance IoU of the refined masks, boxes and scores compared"""
        from cartoonsegmentation_amd import synth
        from cartoonsegmentation_amd.nets import build_isnet, build_rtmdet
        from cartoonsegmentation_amd.weights import SynthWeights
        from oracle import segment as oseg
        t0 = time.perf_counter()
        a = self.pipe.animeinsseg
        S, T = a.default_det_size, a.refine_size
        img = synth.image_u8(self.H, self.W, 5)
        inst = a.infer(img, pred_score_thr=0.3, max_instances=self.INSTANCES, output_type='numpy')
        rp, cfg = build_rtmdet(SynthWeights('rtmdet.'), 1, S, S)
        cfg.max_per_img = self.INSTANCES
        d = oseg.detect(img, rp, cfg, S, pred_score_thr=0.3)
        progs = {}

        def isnet_for(b):
            if b not in progs:
                progs[b] = build_isnet(SynthWeights('isnet.'), b, T, T)
            return progs[b]
        ref = oseg.refine(img, d['masks'], isnet_for, T, 0.3).astype(bool)
        n = min(len(inst), d['n'])
        ious = []
        for k in range(n):
            inter, union = np.logical_and(ref[k], inst.masks[k]).sum(), np.logical_or(ref[k], inst.masks[k]).sum()
            ious.append(float(inter) / float(union) if union else 1.0)
        return {"iou_min": min(ious) if ious else None, "iou_mean": float(np.mean(ious)) if ious else None,
                "instances_gpu": len(inst), "instances_oracle": int(d['n']),
                "masks_bit_identical": bool(n > 0 and len(inst) == d['n'] and np.array_equal(ref, inst.masks)),
                "boxes_and_scores_identical": bool(len(inst) == d['n'] and np.array_equal(d['scores'], inst.scores) and np.array_equal(d['bboxes'], inst.bboxes)),
                "frame": "%dx%d synthetic, det %d, refine %d" % (self.W, self.H, S, T), "seconds": round(time.perf_counter() - t0, 1),
                "what": "AnimeInsSeg.infer (HIP) vs the CPU oracle pipeline on one frame, after threshold; target >= 0.999"}

    # ---- extra single-GPU measurements (batch 1 = BASELINE configs[1..2], n instances in {1, 8}, det 1024, the video) ----
    def _fps(self, batch=None, instances=None, det=None, depth=None, steps=3, conv_roofline=False):
        """frames/s of step() under a variant of the configuration (one untimed step builds + tunes whatever is new)"""
        keep = (self.frames_per_step, self.imgs, self.pipe.max_instances, self.pipe.animeinsseg.default_det_size, self.pipe.cfg.depth_est_size)
        try:
            if batch is not None:
                self.frames_per_step, self.imgs = batch, self.all_imgs[:batch]
            if instances is not None:
                self.pipe.max_instances = instances
            if det is not None:
                self.pipe.animeinsseg.set_detect_size(det)
            if depth is not None:
                self.pipe.cfg.depth_est_size = depth
            self.step(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res = {"frames_per_s": round(steps * self.frames_per_step / dt, 2), "ms_per_frame": round(dt / (steps * self.frames_per_step) * 1e3, 3),
                   "batch": self.frames_per_step, "instances_found": self.n_inst, "det_size": self.pipe.animeinsseg.default_det_size,
                   "depth_size": self.pipe.cfg.depth_est_size}
            if conv_roofline:                                  # the same conv-population roofline as the headline, for this variant
                r = self.roofline()
                res.update(conv_tflops=r["achieved"], conv_frac_of_mfma_peak=r["frac"], conv_avg_launch_us=r["avg_launch_us"])
            return res
        finally:
            self.frames_per_step, self.imgs, self.pipe.max_instances = keep[0], keep[1], keep[2]
            self.pipe.animeinsseg.set_detect_size(keep[3])
            self.pipe.cfg.depth_est_size = keep[4]

    def _video(self):
        """run_kenburns.py on one 1024x1024 image with the shipped yaml's switches (configs/3dkenburns.yaml: inpainting, 75 frames,
        depth_field): 1 seg + 1 depth + autozoom search + 2 inpaint passes + 75 x (warp + bokeh + crop/resize)."""
        pipe = self.pipe
        def once(stages=None):
            t = [time.perf_counter()]
            def mark():
                if stages is not None:
                    torch.cuda.synchronize(); t.append(time.perf_counter())
            kc = pipe.generate_kenburns_config(self.all_imgs[0]); mark()
            kc.depth_field, kc.num_frame = True, 75
            objFrom = {'fltCenterU': kc.int_width / 2.0, 'fltCenterV': kc.int_height / 2.0,
                       'intCropWidth': int(np.floor(0.97 * kc.int_width)), 'intCropHeight': int(np.floor(0.97 * kc.int_height))}
            objTo = pipe.process_autozoom({'fltShift': 100.0, 'fltZoom': 1.25, 'objFrom': objFrom}, kc); mark()
            frames, _ = pipe.process_kenburns({'fltSteps': np.linspace(0.0, 1.0, kc.num_frame).tolist(), 'objFrom': objFrom, 'objTo': objTo,
                                               'boolInpaint': True}, kc, True, False, to_numpy=False)
            torch.cuda.synchronize(); t.append(time.perf_counter())
            if stages is not None:
                stages.update(config_ms=round((t[1] - t[0]) * 1e3, 2), autozoom_ms=round((t[2] - t[1]) * 1e3, 2),
                              inpaint_and_75_frames_ms=round((t[3] - t[2]) * 1e3, 2), points_after_inpaint=int(kc['tenInpaPoints'].shape[2]))
            return len(frames)
        once(); torch.cuda.synchronize()                       # builds the Inpaint programs
        once(); torch.cuda.synchronize()                       # second call: whatever is sized / allocated lazily per video has its steady-state shape now
        st = {}
        once(st)                                               # third call, staged (a synchronize between the stages: the sum is >= video_ms by the bubbles)
        st["stages_sum_ms"] = round(st["config_ms"] + st["autozoom_ms"] + st["inpaint_and_75_frames_ms"], 2)
        t0 = time.perf_counter()
        n = 0
        for _ in range(2):
            n += once()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        st.update(video_ms=round(dt * 1e3, 2), output_frames_per_s=round(75 / dt, 1), videos_per_s=round(1 / dt, 2),
                  what="1 seg + 1 depth + autozoom (256 candidates) + 2 inpaint + 75 x (warp + bokeh + crop/resize), 1024x1024, frames stay on the device")
        return st

    def _warp_points(self):
        """the warp chain by itself (csm_warp_frame_tiled), HIP-event timed: N = P at 1024^2 and the L3-spilling 2048^2 point"""
        from cartoonsegmentation_amd import synth
        out = {}
        for size in (1024, 2048):
            sc = synth.warp_scene(size, size, 1234)
            disp = torch.from_numpy(sc['disp']).to(self.device)
            disp = disp / disp.max() * sc['baseline']
            depth, _, pts, _ = self.ops.disparity_to_points(disp, sc['focal'], sc['baseline'])
            pts, dep, rgb = pts.view(1, 3, -1).contiguous(), depth.view(1, 1, -1).contiguous(), torch.from_numpy(sc['rgb']).to(self.device)
            loc = int(depth.argmin().item())
            settings, common = synth.shift_request(sc, float(depth.min().item()), (loc % size, loc // size))
            shift = self.ops.shift_vector(settings, common)
            for path in ("tiled", "atomics"):
                wf = self.ops.WarpFrame(size, size, self.device, path=path)
                ms = event_time_ms(lambda: wf(pts, rgb, dep, sc['focal'], sc['baseline'], shift), 50, warm=5)
                alg = 155.0 * size * size                       # B_warp = 155 P bytes per frame for N = P, C = 4
                out["%s_%d" % (path, size)] = {"us_per_frame": round(ms * 1e3, 2), "algorithmic_bytes": alg,
                                               "GBps": round(alg / (ms * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            # the same chain with consecutive frames on 3 HIP streams (own scratch each), as KenBurnsPipeline.process_kenburns issues
            # them: frame k + 1's binning runs under frame k's render / hole fill.  Same kernels, same frames; throughput, not latency.
            main = torch.cuda.current_stream(self.device)
            lanes = [(torch.cuda.Stream(self.device), self.ops.WarpFrame(size, size, self.device, path="tiled")) for _ in range(3)]
            def rr(n):
                for st, _ in lanes:
                    st.wait_stream(main)
                for k in range(n):
                    st, wf = lanes[k % 3]
                    with torch.cuda.stream(st):
                        wf(pts, rgb, dep, sc['focal'], sc['baseline'], shift)
                for st, _ in lanes:
                    main.wait_stream(st)
            rr(9); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); rr(60); e1.record(); e1.synchronize()
            ms = e0.elapsed_time(e1) / 60
            alg = 155.0 * size * size
            out["tiled_%d_3streams" % size] = {"us_per_frame": round(ms * 1e3, 2), "GBps": round(alg / (ms * 1e-3) / 1e9, 1),
                                               "frac_of_hbm_peak": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                               "what": "60 frames round-robin over 3 streams with their own scratch (the video loop's issue order)"}
            # ONE call, K frames (csm_warp_frames_tiled): the same overlap inside the library -- internal lanes forked from / joined to the
            # caller's stream by events; the caller issues one asynchronous call on one stream
            wfm = self.ops.WarpFrame(size, size, self.device, path="tiled")
            K = 60 if size == 1024 else 24
            outm = torch.empty((K, size, size, 3), dtype=torch.uint8, device=self.device)
            shifts = [shift] * K
            ms = event_time_ms(lambda: wfm.frames(pts, rgb, dep, sc['focal'], sc['baseline'], shifts, lanes=3, out=outm), 5, warm=2) / K
            out["tiled_%d_multiframe" % size] = {"us_per_frame": round(ms * 1e3, 2), "GBps": round(alg / (ms * 1e-3) / 1e9, 1),
                                                 "frac_of_hbm_peak": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                 "what": "csm_warp_frames_tiled: %d frames per call on ONE caller stream (3 internal lanes)" % K}
            del wfm, outm
            if size == 2048:
                # SURVEY 8(d)'s L3-spilling point: EIGHT DIFFERENT 2048^2 clouds in a batch (8 x 155 P = 5.2 GB of algorithmic traffic, far
                # beyond the 256 MB Infinity Cache: no frame finds its inputs cached), one stream, one scratch
                clouds = []
                for q in range(8):
                    scq = synth.warp_scene(size, size, 2000 + q)
                    dq = torch.from_numpy(scq['disp']).to(self.device)
                    dq = dq / dq.max() * scq['baseline']
                    depq, _, ptq, _ = self.ops.disparity_to_points(dq, scq['focal'], scq['baseline'])
                    clouds.append((ptq.view(1, 3, -1).contiguous(), torch.from_numpy(scq['rgb']).to(self.device), depq.view(1, 1, -1).contiguous()))
                wf8 = self.ops.WarpFrame(size, size, self.device, path="tiled")

                def batch8():
                    for ptq, rgq, dpq in clouds:
                        wf8(ptq, rgq, dpq, sc['focal'], sc['baseline'], shift)
                ms8 = event_time_ms(batch8, 6, warm=2) / 8
                out["tiled_2048_batch8"] = {"us_per_frame": round(ms8 * 1e3, 2), "GBps": round(alg / (ms8 * 1e-3) / 1e9, 1),
                                            "frac_of_hbm_peak": round(alg / (ms8 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                            "what": "8 different 2048x2048 clouds per batch, serial on one stream (inputs never cached: 5.2 GB per batch)"}
                del clouds
        return out

    def variants(self):
        v = {"batch1": self._fps(batch=1, conv_roofline=True), "batch1_lanes3": self._fps_lanes(1, 3, steps=8), "batch1_lanes4": self._fps_lanes(1, 4, steps=8),
             "batch16": self._fps(batch=16, conv_roofline=True), "instances1": self._fps(instances=1),
             "instances8": self._fps(instances=8), "instances100_batch1": self._fps(batch=1, instances=100, steps=2),
             "det1024_batch4": self._fps(batch=4, det=1024), "leres1024_batch4": self._fps(batch=4, depth=1024, steps=2),
             "host_fed": self._fps_host_fed(), "zoe_depth_batch1": self._zoe_variant(), "video": self._video(),
             "warp_chain": self._warp_points()}
        v["reference_shaped_ratio"] = "1 seg + 1 depth + 75 warps: see video (inpaint_and_75_frames_ms vs config_ms)"
        return v

    def weight_buffers(self):
        """the three packed weight buffers, created if this rank's first step did not need them (e.g. no instance -> no ISNet run):
        every rank must take part in the same three broadcasts"""
        import math
        a, pipe = self.pipe.animeinsseg, self.pipe
        nb, S = self.frames_per_step, a.default_det_size
        bufs = [a._detector(S, nb)[1].weights]
        if a.refine_method == 'refinenet_isnet':
            bufs.append(a._refiner(min(a.refine_batch, nb * self.INSTANCES), a.refine_size).weights)
        from cartoonsegmentation_amd.segmentation import scaledown_size
        h, w = scaledown_size(self.H, self.W, pipe.cfg.depth_est_size)
        bufs.append(pipe._leres_prog(int(math.ceil(h / 32) * 32), int(math.ceil(w / 32) * 32), nb).weights)
        # (the packed images of the headline programs: a program of another shape shares them only where its packing is the same --
        # CompiledProgram's `shared` registry -- and would otherwise pack its own copy from the rank's own weight source)
        return bufs

    def broadcast_weights(self):
        """one RCCL broadcast per packed weight buffer (RTMDet 0.37 GB, ISNet 0.18 GB, LeReS 0.63 GB), start-up only"""
        from cartoonsegmentation_amd import shard
        n = 0
        for w in self.weight_buffers():
            shard.broadcast_weights(w, self.dist, src=0)
            n += w.numel() * 4
        return n

    def _programs(self):
        """every compiled layer program of the pipeline with random ext tensors of the right shapes"""
        a = self.pipe.animeinsseg
        cps = [("rtmdet-ins-l@%d n=%d" % k, v[1]) for k, v in a._det_programs.items()]
        cps += [("isnet n=%d@%d" % k, v) for k, v in a._refine_programs.items()]
        cps += [("leres@%dx%d n=%d%s" % (k[1], k[0], k[2], "" if k[3] == 0 else " #%d" % k[3]), v) for k, v in self.pipe._leres.items()]
        items = []
        for name, cp in cps:
            ext = sorted((b for b in cp.prog.bufs if b.ext >= 0), key=lambda b: b.ext)
            items.append((name, cp, [torch.randn(b.n, b.c, b.h, b.w, device=self.device) for b in ext]))
        return items

    def config(self, world):
        return {"workload": "seg(RTMDet-Ins-L det640 + maskhead + ISNet refine@720, %d instances) + LeReS depth@640 + 1 warp, "
                            "frame %dx%d" % (self.n_inst or self.INSTANCES, self.W, self.H),
                "frames_per_gpu_step": self.frames_per_step, "global_batch": self.frames_per_step * world,
                "is_baseline_configs3": bool(world == 8 and self.frames_per_step * world == 64 and self.W == 1024),
                "parallelism": "frames sharded x%d (no data-path collective; per-rank gather of uint8 frames + bit-packed instance masks)" % world,
                "weights": "closed-form synthetic (no checkpoints offline)", "precision": "fp32 exact (v_mfma_f32_32x32x2_f32)"}

    def roofline(self):
        """dominant kernel = the implicit-GEMM conv (k_conv_dma tiles + k_conv_mfma fallback): algorithmic conv FLOPs / summed launch
        durations (HIP events around every op on the launch stream)"""
        tot_ms, tot_fl, tot_ex, wino_ms, n_wino, per_net, per_class = 0.0, 0.0, 0.0, 0.0, 0, {}, {}
        progs = self._programs()
        before = [cp.runs for _, cp, _ in progs]
        self.step(); torch.cuda.synchronize()                      # how often each program runs in one step (ISNet: once per 8 instances)
        n_launch, alg_bytes = 0, 0
        for (name, cp, ext), b in zip(progs, before):
            per_step = cp.runs - b
            if per_step == 0:                                      # a program of another variant (other batch / det size): not part of this step
                continue
            cp.run(*ext)
            ms = None
            for _ in range(3):
                m = cp.profile(*ext)
                ms = m if ms is None else [min(a, b) for a, b in zip(ms, m)]
            cms = sum(t for t, o in zip(ms, cp.prog.ops) if o['kind'] == 1)
            nconv = sum(1 for o in cp.prog.ops if o['kind'] == 1)
            wms = sum(t for t, o in zip(ms, cp.prog.ops) if o['kind'] == 1 and o['flags'] & 12)      # Winograd layers: F(2x2) k_conv_wino8 (flag 4), F(4x4) k_conv_wino4 (flag 8)
            nw = sum(1 for o in cp.prog.ops if o['kind'] == 1 and o['flags'] & 12)
            for t, o in zip(ms, cp.prog.ops):                     # per-class table: where the step's conv time goes and at what rate
                if o['kind'] == 1:
                    c = per_class.setdefault(conv_op_class(cp.prog, o), [0, 0.0, 0.0, 0.0])
                    fn, fe = conv_op_flops(cp.prog, o)
                    c[0] += per_step; c[1] += t * per_step; c[2] += fe * per_step; c[3] += fn * per_step
            per_net[name] = {"conv_ms": round(cms, 3), "all_ops_ms": round(sum(ms), 3), "gflop": round(cp.prog.flops / 1e9, 1),
                             "gflop_executed": round(cp.prog.flops_exec / 1e9, 1), "conv_launches": nconv, "winograd_launches": nw,
                             "winograd_ms": round(wms, 3), "runs_per_step": per_step}
            tot_ms += cms * per_step; tot_fl += cp.prog.flops * per_step; tot_ex += cp.prog.flops_exec * per_step; n_launch += nconv * per_step
            wino_ms += wms * per_step; n_wino += nw * per_step
            for o in cp.prog.ops:                                  # algorithmic bytes of a conv = input + output + weights, once each
                if o['kind'] == 1:
                    vi, vo, nat = cp.prog.views[o['in0']], cp.prog.views[o['out']], o['nat']
                    alg_bytes += per_step * 4 * (vi.n * vi.h * vi.w * vi.c + vo.n * vo.h * vo.w * vo.c +
                                                 nat['cout_g'] * nat['groups'] * nat['cin_g'] * o['kh'] * o['kw'])
        # `achieved` / `frac` price the matrix pipe on the FLOPs it EXECUTES: a Winograd F(2x2, 3x3) layer executes 16 / 36 of its natural
        # (direct-convolution) multiply-adds, so natural FLOPs / time would read as skipped work.  The natural-FLOP rate is reported beside it.
        ach = tot_ex / (tot_ms * 1e-3) / 1e12
        ach_nat = tot_fl / (tot_ms * 1e-3) / 1e12
        tot_fl /= self.frames_per_step
        tot_ex /= self.frames_per_step
        return {"bound": "mfma", "kernel": "k_conv_dma+k_conv_wino4+k_conv_wino8+k_conv_mfma+k_conv_grouped (fp32 implicit GEMM / Winograd F(4x4,3x3) and F(2x2,3x3) / vector-pipe narrow groups, all conv launches of one step = %d frames)" % self.frames_per_step,
                "vendor_fp32_gemm_context": self._vendor_gemm_context() if self.frames_per_step == 8 else None,
                "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TF, 4),
                "flops_basis": "executed MFMA FLOPs (Winograd layers: F(4x4) 36 products per 4x4 output tile and channel pair, F(2x2) 16 per 2x2)",
                "direct_equivalent_tflops": round(ach_nat, 2), "direct_equivalent_frac": round(ach_nat / MFMA_F32_PEAK_TF, 4),
                "executed_flops_per_frame": tot_ex, "winograd_launches_per_step": n_wino, "winograd_ms_per_step": round(wino_ms, 3),
                "traffic": load_traffic("k_conv"), "traffic_source": TRAFFIC_SOURCE,
                "algorithmic_bytes_per_launch": int(alg_bytes / max(n_launch, 1)),
                "algorithmic_flops_per_frame": tot_fl,
                "avg_launch_us": round(tot_ms * 1e3 / max(n_launch, 1), 2), "launches_per_step": n_launch, "conv_ms_per_step": round(tot_ms, 3),
                "note": "a launch = one conv op of a layer program (a mixed-tile or split-K op issues two kernels: rocprofv3's per-kernel "
                        "average is lower, its k_conv_* total per step is the comparable figure -- profiles/README.md)", "per_net": per_net,
                "per_class": {k: {"launches_per_step": v[0], "ms_per_step": round(v[1], 3), "executed_tflops": round(v[2] / v[1] / 1e9, 1),
                                  "frac_of_mfma_peak": round(v[2] / v[1] / 1e9 / MFMA_F32_PEAK_TF, 4), "direct_equivalent_tflops": round(v[3] / v[1] / 1e9, 1)}
                              for k, v in sorted(per_class.items(), key=lambda kv: -kv[1][1]) if v[1] > 0}}

    def _vendor_gemm_context(self):
        """context for `frac`, measured in this run: what the vendor library's fp32 GEMM (torch.mm -> hipBLASLt / rocBLAS, TF32 off, the
        im2col matrix handed over for free) reaches on the GEMM shapes of the layers that carry the conv time at 8 frames per step.  Not part of
        any product path; see profiles/r04_vendor_gemm_ceiling.txt for the same table next to this build's conv ops."""
        prev = torch.backends.cuda.matmul.allow_tf32
        torch.backends.cuda.matmul.allow_tf32 = False
        res = {}
        try:
            for name, M, N, K in (("leres_1x1_1024_1024_at_40", 12800, 1024, 1024), ("leres_3x3_256_256_at_160", 204800, 256, 2304),
                                  ("isnet_3x3_64_64_at_360", 2073600, 64, 576), ("rtmdet_3x3_256_256_at_80", 51200, 256, 2304)):
                a_, b_ = torch.randn(M, K, device=self.device), torch.randn(K, N, device=self.device)
                ms = event_time_ms(lambda: a_ @ b_, 5, warm=2)
                res[name] = {"M_N_K": [M, N, K], "tflops": round(2.0 * M * N * K / ms / 1e9, 1), "frac_of_mfma_peak": round(2.0 * M * N * K / ms / 1e9 / MFMA_F32_PEAK_TF, 3)}
                del a_, b_
        finally:
            torch.backends.cuda.matmul.allow_tf32 = prev
        torch.cuda.empty_cache()
        return res

    def extra(self):
        return {}

    def cpu_baseline(self, seconds):
        from oracle import frame as oframe
        return oframe.cpu_baseline(seconds, frame=self.H, instances=self.INSTANCES)


def make_workload(kind, size, rank, device, world, dist, batch=8):
    if kind in ("auto", "frame"):
        wl = FrameWorkload(size, rank, device, batch)
    elif kind == "warp":
        wl = WarpWorkload(size, rank, device)
    else:
        raise SystemExit("unknown workload %r" % kind)
    wl.attach(world, dist, device)
    return wl


TRAFFIC_SOURCE = ("profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed by the builder; "
                  "NOT re-measured inside this run (PMC collection needs its own rocprofv3 passes)")


def conv_op_flops(prog, o):
    """(natural, executed) FLOPs of one CONV op of a lowered program: a Winograd layer executes 16 products per 2x2 (F(2x2), flag 4) or 36
    per 4x4 (F(4x4), flag 8) output tile and channel pair instead of 9 per pixel"""
    vo, nat = prog.views[o['out']], o['nat']
    cc = nat['cout_g'] * nat['groups'] * nat['cin_g']
    natural = 2.0 * vo.n * vo.h * vo.w * cc * o['kh'] * o['kw']
    if o['flags'] & 8:
        return natural, 2.0 * vo.n * ((vo.h + 3) // 4) * ((vo.w + 3) // 4) * 36 * cc
    if o['flags'] & 4:
        return natural, 2.0 * vo.n * ((vo.h + 1) // 2) * ((vo.w + 1) // 2) * 16 * cc
    return natural, natural


def conv_op_class(prog, o):
    """coarse class of a CONV op for the per-class roofline table"""
    vo, nat = prog.views[o['out']], o['nat']
    if o['flags'] & 8:
        return "winograd_f4x4"
    if o['flags'] & 4:
        return "winograd_f2x2"
    if o['flags'] & 2:
        return "stem"
    if nat['groups'] > 1:
        return "grouped"
    if o['kh'] == 1 and o['kw'] == 1:
        return "pointwise_map_ge_80x80" if vo.h * vo.w >= 6400 else "pointwise_map_lt_80x80"
    return "direct_kxk_map_ge_80x80" if vo.h * vo.w >= 6400 else "direct_kxk_map_lt_80x80"


def load_traffic(kernel_key):
    """HBM bytes per launch from the committed rocprofv3 --pmc pass (profiles/traffic.json), or None"""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get(kernel_key)
        except Exception:
            return None
    return None


def main():
    a = parse()
    if os.environ.get("CSM_BENCH_WATCHDOG"):          # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["CSM_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    # CSM_BENCH_BACKEND=gloo + fewer GPUs than ranks is a logic check of the multi-rank path on a 1-GPU box (ranks share a GPU);
    # the measured configuration is always one rank per GPU over RCCL ("nccl").
    backend = os.environ.get("CSM_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    # CSM_BENCH_FORCE_DIST=1: take the process-group path with ONE rank as well (a 1-GPU box then runs the whole multi-rank start-up --
    # init_process_group("nccl", device_id=...), the tile-table / weight broadcasts, the all-reduces and the asynchronous gather -- on RCCL itself)
    if world > 1 or os.environ.get("CSM_BENCH_FORCE_DIST", "0") not in ("", "0"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if dist is not None:
        # multi-rank start-up: rank 0 builds the weights and tunes the conv tiles, the others wait, then build with the tile table
        # rank 0 saved (CSM_TUNE_CACHE: no tuning launches on 7 of 8 ranks) and with PLACEHOLDER weights (zeros of the right shapes):
        # what they compute with arrives through the RCCL broadcast below, like a checkpoint read by rank 0 would
        # (the table travels through the process group -- broadcast_object_list -- not through a shared file: every rank keeps a PRIVATE copy)
        # the file is private to THIS process (pid in the name, removed at exit): a table left behind by another run -- another build of the
        # library, another port reuse -- can never be picked up
        import atexit
        import tempfile
        tune_path = os.path.join(tempfile.gettempdir(), "csm_tune_%s_%s_rank%d_pid%d.txt" % (os.environ.get("MASTER_PORT", "0"), a.size, rank, os.getpid()))
        os.environ["CSM_TUNE_CACHE"] = tune_path
        if os.path.exists(tune_path):
            os.remove(tune_path)
        atexit.register(lambda: os.path.exists(tune_path) and os.remove(tune_path))
        if rank != 0:
            os.environ["CSM_WEIGHTS_PLACEHOLDER"] = "1"
    wl = make_workload(a.workload, a.size, rank, device, world, dist, a.batch)

    weights_equal = None
    if dist is None:
        wl.step_and_gather()                  # compiles the layer programs (lazy) -- untimed
        bcast_bytes = 0
    else:
        if rank == 0:
            wl.step(); torch.cuda.synchronize()       # builds + tunes + saves the tile table -- untimed
        table = [open(os.environ["CSM_TUNE_CACHE"]).read() if rank == 0 and os.path.exists(os.environ["CSM_TUNE_CACHE"]) else None]
        dist.broadcast_object_list(table, src=0)      # rank 0's tuned tiles: 7 of 8 ranks time nothing and all ranks run the same tiles
        if rank != 0 and table[0]:
            with open(os.environ["CSM_TUNE_CACHE"], "w") as f:
                f.write(table[0])
        tile_table_lines = len(table[0].splitlines()) if table[0] else 0
        dist.barrier()
        if rank != 0:
            wl.step(); torch.cuda.synchronize()       # same programs, tiles from the table, placeholder weights -- untimed
        dist.barrier()
        bcast_bytes = wl.broadcast_weights() if hasattr(wl, "broadcast_weights") else 0
        if hasattr(wl, "weight_buffers"):             # every rank must now hold rank 0's weights bit for bit
            sums = torch.stack([w.double().sum() for w in wl.weight_buffers()])
            hi, lo = sums.clone(), sums.clone()
            dist.all_reduce(hi, op=dist.ReduceOp.MAX); dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            weights_equal = bool(torch.equal(hi, lo)) and bool((hi != 0).all())
            assert weights_equal, "weights differ between ranks after the broadcast"
        wl.step_and_gather()                  # ranks != 0 now hold real weights: build whatever their first step skipped -- untimed
    for _ in range(a.warmup):
        wl.step_and_gather()
    wl.finish_gathers()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        wl.step_and_gather()
    wl.finish_gathers()                                # every step's frames are on rank 0 before the clock stops
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        frames = a.steps * world * wl.frames_per_step
        out = {"metric": "frames/sec at 1024x1024 (seg+depth+warp)", "value": round(frames / dt, 3),
               "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": wl.dtype, "data": "synthetic",
               "config": wl.config(world)}
        out["roofline"] = None if a.no_roofline else wl.roofline()
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = wl.cpu_baseline(a.cpu_seconds)
        if world == 1 and not a.no_iou and hasattr(wl, "mask_iou_vs_oracle"):
            out["mask_iou_vs_oracle"] = wl.mask_iou_vs_oracle()    # BASELINE.json's `mask IoU vs ref`; the oracle is the checker, untimed
        out.update(wl.extra())
        if world == 1 and not a.no_variants and hasattr(wl, "variants"):
            out["variants"] = wl.variants()
            b1 = out["variants"].get("batch1")
            if b1:             # the literal BASELINE configs[1..2] (one frame per step) next to the headline (configs[3]'s 8 frames per rank)
                out["batch1"] = {"frames_per_s": b1["frames_per_s"], "ms_per_frame": b1["ms_per_frame"],
                                 "conv_frac_of_mfma_peak": b1.get("conv_frac_of_mfma_peak"),
                                 "frames_per_s_3_in_flight": out["variants"].get("batch1_lanes3", {}).get("frames_per_s"),
                                 "frames_per_s_4_in_flight": out["variants"].get("batch1_lanes4", {}).get("frames_per_s"),
                                 "what": "BASELINE configs[1..2] literally: ONE 1024x1024 frame per step (seg + depth + warp), serial loop; "
                                         "*_in_flight: the same single-frame steps with 3 / 4 frames in flight (FrameLanes)"}
        if dist is not None:
            out["process_group"] = {"backend": dist.get_backend(), "world_size": world}
            out["gather"] = wl.check_gathered()
            out["weights_broadcast_bytes"] = bcast_bytes
            out["weights_equal_after_broadcast"] = weights_equal
            out["tile_table"] = {"entries": tile_table_lines, "how": "tuned by rank 0, broadcast_object_list to the other ranks"}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
