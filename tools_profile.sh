#!/bin/bash
# usage: tools_profile.sh <tag> [quick] -- GPU validation + measurement pass on a gpurun box; writes gpurun_out/<tag>/
# every rocprofv3 call: --output-format csv (the default rocpd output stalled for minutes here) and a hard timeout
TAG=${1:-run}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
export CSM_TUNE_CACHE=/tmp/csm_tiles.txt   # first bench run tunes + saves; the profiled runs reuse the tiles (no tuning launches)
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $OUT/pytest_gpu.log; tail -2 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/smoke.log
T0=$(date +%s); timeout 900 python bench.py 2>/dev/null | tail -1 > $OUT/bench_frame.json; head -c 400 $OUT/bench_frame.json; echo
echo "python bench.py (default flags, all variants): $(( $(date +%s) - T0 )) s wall" | tee $OUT/bench_seconds.txt
timeout 300 python bench.py --workload warp --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_warp.json; head -c 300 $OUT/bench_warp.json; echo
timeout 300 python tools/layer_profile.py 2>/dev/null > $OUT/layer_profile.txt; grep "^==" $OUT/layer_profile.txt
LP_BATCH=8 timeout 300 python tools/layer_profile.py 2>/dev/null > $OUT/layer_profile_b8.txt; grep "^==" $OUT/layer_profile_b8.txt
timeout 200 python tools/video_breakdown.py 2>/dev/null | grep rep > $OUT/video_breakdown.txt; cat $OUT/video_breakdown.txt
timeout 100 python tools/time_autozoom.py 1024 2>/dev/null | grep autozoom | head -1 > $OUT/autozoom.txt; cat $OUT/autozoom.txt
(timeout 200 python tools/zoe_core_profile.py 672 672; timeout 200 python tools/zoe_core_profile.py 384 512) 2>/dev/null > $OUT/zoe_core_profile.txt; cat $OUT/zoe_core_profile.txt
python tools/check_isa_barriers.py > $OUT/isa_barriers.txt 2>&1; cat $OUT/isa_barriers.txt
[ "$2" = "quick" ] && exit 0
cd /tmp && export TMPDIR=/tmp
RP="timeout 200 rocprofv3 --kernel-trace --stats --output-format csv"
# --no-roofline: the process then executes exactly 1 + warmup + steps = 9 steps (no per-op event pass), so totals divide cleanly
$RP -d $OUT/stats -o frame -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-variants --no-roofline > $OUT/stats.log 2>&1
# same workload with the LeReS side stream off: kernels do not overlap, so durations are per-kernel clean (HIP-event comparable)
CSM_OVERLAP_DEPTH=0 $RP -d $OUT/stats_serial -o frame -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-variants --no-roofline > $OUT/stats_serial.log 2>&1
$RP -d $OUT/stats_warp -o warp -- python /root/repo/bench.py --workload warp --steps 100 --warmup 10 --no-cpu-baseline > $OUT/stats_warp.log 2>&1
CSM_WARP_PATH=atomics $RP -d $OUT/stats_warp_atomics -o warp -- python /root/repo/bench.py --workload warp --steps 100 --warmup 10 --no-cpu-baseline > $OUT/stats_warp_atomics.log 2>&1
[ "$2" = "full" ] && $RP -d $OUT/stats_video -o video -- python /root/repo/tools/video_breakdown.py > $OUT/stats_video.log 2>&1
[ "$2" = "full" ] && $RP -d $OUT/stats_autozoom -o az -- python /root/repo/tools/time_autozoom.py 1024 > $OUT/stats_autozoom.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o frame -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-roofline > $OUT/pmc_$C.log 2>&1
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmcw_$C -o warp -- python /root/repo/bench.py --workload warp --steps 20 --warmup 2 --no-cpu-baseline > $OUT/pmcw_$C.log 2>&1
done
python - <<PY
import csv,glob,collections,json
for f in sorted(glob.glob("$OUT/stats*/**/*kernel_stats.csv", recursive=True)):
    print(f); print("".join(l[:170]+"\n" for l in open(f).readlines()[:12]))
    calls=tot=0
    for r in csv.DictReader(open(f)):
        if "k_conv_" in r["Name"] or "k_splitk_reduce" in r["Name"]: calls+=int(r["Calls"]); tot+=int(r["TotalDurationNs"])   # the reduce kernel belongs to its conv op
    if calls:
        msg="ALL k_conv_* kernel launches: calls %d total %.3f ms average %.2f us"%(calls,tot/1e6,tot/calls/1e3)
        if "/stats/" in f or "/stats_serial/" in f:      # bench.py --steps 6 --warmup 2 --no-roofline = 9 executed steps
            ops=json.load(open("$OUT/bench_frame.json"))["roofline"]["launches_per_step"]
            msg+="; 9 steps -> %.3f ms of conv kernels per step = %.2f us per conv op (%d ops per step; bench.py's HIP-event figure: roofline.avg_launch_us)"%(tot/1e6/9,tot/9/ops/1e3,ops)
        print(msg); open(f.replace("kernel_stats.csv","conv_summary.txt"),"w").write(msg+"\n")
tr={}
for kind,key in (("FETCH_SIZE","fetch_KB"),("WRITE_SIZE","write_KB")):
    for pre in ("pmc_","pmcw_"):
        agg=collections.defaultdict(lambda:[0,0.0])
        for f in glob.glob("$OUT/%s%s/**/*counter_collection.csv"%(pre,kind), recursive=True):
            for r in csv.DictReader(open(f)):
                k=r["Kernel_Name"]; k=k[k.find("k_"):][:24] if "k_" in k else k[:24]
                k=k.split("(")[0].split("<")[0]
                if k.startswith("k_conv_") or k.startswith("k_splitk_reduce"): k="k_conv"            # all conv tile instantiations (+ split-K reduce) together
                agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
        for k,(n,v) in agg.items():
            tr.setdefault(k,{})[key]=round(v/n,1); tr[k]["launches_"+key]=n
json.dump(tr, open("$OUT/pmc_summary.json","w"), indent=1)
# HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE tallies 128-B requests at 64 B -- MI355X_MICROARCH.md, HBM)
traffic={k:int(2*v.get("fetch_KB",0)*1024+v.get("write_KB",0)*1024) for k,v in tr.items() if k.startswith("k_")}
# conv: per OP of the layer programs (a mixed-tile / split-K op issues two kernels): total over the run / (4 executed steps x ops per step)
if "k_conv" in tr:
    ops=json.load(open("$OUT/bench_frame.json"))["roofline"]["launches_per_step"]
    tot=2*tr["k_conv"].get("fetch_KB",0)*1024*tr["k_conv"].get("launches_fetch_KB",0)+tr["k_conv"].get("write_KB",0)*1024*tr["k_conv"].get("launches_write_KB",0)
    traffic["k_conv_per_kernel_launch"]=traffic["k_conv"]; traffic["k_conv"]=int(tot/4/ops); traffic["k_conv_per_step"]=int(tot/4)
chain=[k for k in ("k_tile_bin","k_tile_render","k_tile_holes") if k in traffic]
if chain: traffic["warp_chain_tiled"]=sum(traffic[k] for k in chain)
traffic["_note"]="HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE counts 128-B requests as 64 B, MI355X_MICROARCH.md; WRITE_SIZE uncalibrated); separate --pmc passes; k_conv = per conv OP of the layer programs: bytes of all k_conv_* kernels of the run / (4 executed steps of bench.py --steps 2 --warmup 1 --no-roofline x conv ops per step), frame workload at batch 8; warp_chain_tiled = sum over the three kernels (k_tile_bin, k_tile_render, k_tile_holes) of one csm_warp_frame_tiled call"
json.dump(traffic, open("$OUT/traffic.json","w"), indent=1)
for k,v in sorted(tr.items(), key=lambda kv:-kv[1].get("fetch_KB",0))[:16]: print(k, v)
PY
# pipe-utilisation counters (SQ / LDS / L2) of the kernels that carry the conv time + the warp render pass: one table
# (the Winograd layers -- round 6: k_conv_wino4 -- + two direct layers + the warp render pass; tools/gpu/pmc_layers.sh)
bash /root/repo/tools/gpu/pmc_layers.sh > $OUT/pmc_sq.log 2>&1; python /root/repo/tools/pmc_table.py /root/repo/gpurun_out/r06pmc/summary.txt > $OUT/conv_pmc.txt 2>&1; cat $OUT/conv_pmc.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +1M -delete; find $OUT -name "*agent_info.csv" -delete
