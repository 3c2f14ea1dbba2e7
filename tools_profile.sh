#!/bin/bash
# usage: tools_profile.sh <tag> -- runs tests, bench, rocprofv3 stats + pmc passes; writes gpurun_out/<tag>/
TAG=${1:-run}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log
python bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 > $OUT/bench.json; cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o warp -- python /root/repo/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o warp -- python /root/repo/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o warp -- python /root/repo/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head; 
python - <<PY
import csv,glob,collections
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:3000])
for kind in ("fetch","write"):
    agg=collections.defaultdict(lambda:[0,0.0])
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv"%kind, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:60]; agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
    for k,(n,v) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:12]:
        print(kind, k, n, "avg_KB=%.1f"%(v/n))
PY
# drop the big raw traces, keep summaries
find $OUT -name "*kernel_trace.csv" -size +2M -delete
