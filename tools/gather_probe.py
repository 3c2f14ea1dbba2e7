#!/usr/bin/env python3
"""where the per-step cost of the multi-rank output gather goes (bench.py Workload.step_and_gather), on one rank over RCCL:
times 16 headline steps (8 frames) in one process as (a) step only, (b) step + staging copy, (c) step + staging copy + asynchronous gather."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
import torch
import torch.distributed as dist
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
wl = bench.make_workload("frame", 1024, 0, dev, 1, None, 8)
wl.step(); torch.cuda.synchronize()


def plain(n=16, warm=3):
    for _ in range(warm):
        wl.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        wl.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("before init_process_group: step %.2f ms, %.2f ms" % (plain(), plain()), flush=True)
if os.environ.get("PROBE_DEVICE_ID", "1") == "1":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
else:
    dist.init_process_group("nccl", rank=0, world_size=1)
print("after init_process_group (no collective yet): step %.2f ms, %.2f ms" % (plain(), plain()), flush=True)
t = torch.ones(4, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
print("after the first collective: step %.2f ms, %.2f ms" % (plain(), plain()), flush=True)
wl.attach(1, dist, dev)
wl.step_and_gather(); wl.finish_gathers(); torch.cuda.synchronize()


def timed(fn, n=16, warm=3):
    for _ in range(warm):
        fn()
    wl.finish_gathers(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    wl.finish_gathers(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


stage = torch.empty((wl.frames_per_step, wl.rb), dtype=torch.uint8, device=dev)


def step_copy():
    wl.step()
    stage.copy_(wl.records[:wl.frames_per_step])


def step_gather_sync():
    wl.step()
    stage.copy_(wl.records[:wl.frames_per_step])
    dist.gather(stage, wl.gather_list, dst=0)


for rep in range(2):
    print("rep %d: step only %.2f ms | + staging copy %.2f | + async gather (bench path) %.2f | + blocking gather %.2f"
          % (rep, timed(wl.step), timed(step_copy), timed(wl.step_and_gather), timed(step_gather_sync)), flush=True)
dist.destroy_process_group()
wl.dist = None
print("after destroy_process_group: step %.2f ms, %.2f ms" % (plain(), plain()), flush=True)
