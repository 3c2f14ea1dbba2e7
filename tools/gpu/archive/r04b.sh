#!/bin/bash
# round 4, GPU call B: pipe-utilisation counters (SQ / LDS / L2) for the kernels that carry the conv time now, one counter set per pass
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp CSM_AUTOTUNE=0
rocprofv3 -L > $O/counters_avail.txt 2>&1
# layer (n h w cin cout k) and forced tile configuration
LAYERS=("8 160 160 256 256 3 45" "8 320 320 256 128 3 50" "8 40 40 1024 1024 1 6" "16 360 360 64 64 3 44" "8 40 40 256 256 3 47" "8 80 80 256 256 3 44" "8 160 160 256 256 1 41" "16 360 360 32 64 3 52")
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC" "FETCH_SIZE" "WRITE_SIZE")
for L in "${LAYERS[@]}"; do
  tag=$(echo $L | tr ' ' '_')
  for C in "${SETS[@]}"; do
    c1=$(echo $C | cut -d' ' -f1)
    timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/${tag}_$c1 -o l -- python $R/tools/pmc_layer.py $L > $O/${tag}_$c1.log 2>&1
  done
done
# the warp chain's render pass under the same counter sets (tag warp_1024)
for C in "${SETS[@]}"; do
  c1=$(echo $C | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/warp_1024_$c1 -o l -- python $R/bench.py --workload warp --steps 20 --warmup 2 --no-cpu-baseline > $O/warp_1024_$c1.log 2>&1
done
python - <<PY
import csv,glob,collections,os
out=open("$O/summary.txt","w")
for d in sorted(glob.glob("$O/*/")):
    agg=collections.defaultdict(lambda:[0,0.0]); names=set()
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_conv" in r["Kernel_Name"] or ("warp_1024" in d and "k_tile_render" in r["Kernel_Name"]):
                a=agg[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"]); names.add(r["Kernel_Name"][:70])
    line="%s %s %s" % (d.rstrip("/").split("/")[-1], {k:"%.4g"%(v[1]/v[0]) for k,v in agg.items()}, sorted(names)[:2])
    print(line); out.write(line+"\n")
PY
# keep only the summaries (the raw csv trees are large)
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
