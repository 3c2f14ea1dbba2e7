#!/bin/bash
# usage: tools_profile.sh <tag> [quick] -- GPU validation + measurement pass; writes gpurun_out/<tag>/
TAG=${1:-run}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
export CSM_TUNE_CACHE=/tmp/csm_tiles.txt   # first bench run tunes + saves; the profiled runs reuse the tiles (no tuning launches)
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/smoke.log
python bench.py 2>/dev/null | tail -1 > $OUT/bench.json; cat $OUT/bench.json
python bench.py --workload warp --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_warp.json; cat $OUT/bench_warp.json
python tools/conv_bench.py --sweep 2>/dev/null > $OUT/conv_sweep.txt; tail -1 $OUT/conv_sweep.txt
python tools/layer_profile.py 2>/dev/null > $OUT/layer_profile.txt; grep "^==" $OUT/layer_profile.txt
[ "$2" = "quick" ] && exit 0
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o frame -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1
# same workload with the LeReS side stream off: kernels do not overlap, so durations are per-kernel clean (HIP-event comparable)
CSM_OVERLAP_DEPTH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -o frame -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/stats_serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_warp -o warp -- python /root/repo/bench.py --workload warp --steps 100 --warmup 10 --no-cpu-baseline > $OUT/stats_warp.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o frame -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$C.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmcw_$C -o warp -- python /root/repo/bench.py --workload warp --steps 20 --warmup 2 --no-cpu-baseline > $OUT/pmcw_$C.log 2>&1
done
python - <<PY
import csv,glob,collections,json
out={}
for f in glob.glob("$OUT/stats*/**/*kernel_stats.csv", recursive=True):
    print(f); print("".join(l[:170]+"\n" for l in open(f).readlines()[:14]))
    calls=tot=0
    for r in csv.DictReader(open(f)):
        if "k_conv_" in r["Name"]: calls+=int(r["Calls"]); tot+=int(r["TotalDurationNs"])
    if calls:
        msg="ALL k_conv_* launches: calls %d total %.3f ms average %.2f us"%(calls,tot/1e6,tot/calls/1e3)
        print(msg); open(f.replace("kernel_stats.csv","conv_summary.txt"),"w").write(msg+"\n")
tr={}
for kind,key in (("FETCH_SIZE","fetch_KB"),("WRITE_SIZE","write_KB")):
    for pre in ("pmc_","pmcw_"):
        agg=collections.defaultdict(lambda:[0,0.0])
        for f in glob.glob("$OUT/%s%s/**/*counter_collection.csv"%(pre,kind), recursive=True):
            for r in csv.DictReader(open(f)):
                k=r["Kernel_Name"]; k=k[k.find("k_"):][:24] if "k_" in k else k[:24]
                if k.startswith("k_conv_"): k="k_conv"            # all conv tile instantiations together
                agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
        for k,(n,v) in agg.items():
            tr.setdefault(k,{})[key]=round(v/n,1); tr[k]["launches_"+key]=n
json.dump(tr, open("$OUT/pmc_summary.json","w"), indent=1)
# HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE tallies 128-B requests at 64 B -- MI355X_MICROARCH.md, HBM)
traffic={k:int(2*v.get("fetch_KB",0)*1024+v.get("write_KB",0)*1024) for k,v in tr.items() if k.startswith("k_")}
for k in list(traffic):
    if k.startswith("k_update_output"): traffic["k_update_output"]=traffic[k]
traffic["_note"]="HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE counts 128-B requests as 64 B, MI355X_MICROARCH.md; WRITE_SIZE uncalibrated); separate --pmc passes; k_conv = launch-weighted average over all conv launches (k_conv_dma tiles + k_conv_mfma) of the frame workload"
json.dump(traffic, open("$OUT/traffic.json","w"), indent=1)
for k,v in sorted(tr.items(), key=lambda kv:-kv[1].get("fetch_KB",0))[:14]: print(k, v)
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +1M -delete
