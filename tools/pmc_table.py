#!/usr/bin/env python3
"""derived pipe-utilisation table from the per-pass rocprofv3 --pmc summaries of tools/gpu/r04b.sh (gpurun_out/r04b/summary.txt):
one row per (layer, tile configuration).  Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles,
SQ_VALU_MFMA_BUSY_CYCLES cycles (64 per v_mfma_f32_32x32x2_f32), GRBM_GUI_ACTIVE is summed over the 8 XCDs, FETCH_SIZE / WRITE_SIZE are KB
and FETCH_SIZE reports half of a wide streaming read on gfx950 (doubled here)."""
import ast, collections, re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r04b/summary.txt"
rows = collections.OrderedDict()
for line in open(path):
    m = re.match(r"(\d+(?:_\d+){6}|warp_1024)_(\w+) (\{.*?\}) (\[.*\])", line.strip())
    if not m:
        continue
    d = rows.setdefault(m.group(1), {"kernel": ast.literal_eval(m.group(4))[0] if ast.literal_eval(m.group(4)) else "?"})
    d.update({k: float(v) for k, v in ast.literal_eval(m.group(3)).items()})
print("%-26s %-44s %8s %6s %6s %6s %6s %6s %7s %7s %6s %8s %8s %8s" % ("layer n_h_w_cin_cout_k_cfg", "kernel", "us", "MFMA%", "w/SIMD", "park%", "stall%", "issue%", "LDSbusy", "LDSconf", "L2hit", "fetchMB", "writeMB", "algMB"))
for key, d in rows.items():
    cyc = d["GRBM_GUI_ACTIVE"] / 8.0
    us = cyc / 2400.0
    if key == "warp_1024":
        alg = 0.0                                  # k_tile_render alone: SURVEY's 155 P bytes are for the whole chain
    else:
        n, h, w, cin, cout, k, cfg = [int(v) for v in key.split("_")]
        alg = 4e-6 * (n * h * w * cin + n * h * w * cout + cout * cin * k * k)
    kern = re.sub(r"void \(anonymous namespace\)::", "", d["kernel"])[:44]
    print("%-26s %-44s %8.1f %6.1f %6.2f %6.1f %6.1f %6.1f %7.1f %7.1f %6.1f %8.1f %8.1f %8.1f" % (
        key, kern, us, 100 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 4 * d["SQ_WAVE_CYCLES"] / (1024 * cyc),
        100 * d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 100 * d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], 100 * d["SQ_ACTIVE_INST_ANY"] / d["SQ_WAVE_CYCLES"],
        100 * d["SQ_LDS_IDX_ACTIVE"] / (256 * cyc), 100 * d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1),
        100 * d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 2 * d["FETCH_SIZE"] / 1e3, d["WRITE_SIZE"] / 1e3, alg))
