"""Fold the per-pass CSVs of tools/profile_pmc.sh (gpurun_out/<tag>_pmc/<workload>_<pass>.csv; tag = argv[1], default r03) into
   profiles/<tag>_pmc_<workload>.csv  : one row per (kernel, grid, counter) for OUR kernels (k_*), as measured
   profiles/<tag>_pmc_summary.csv     : one row per (workload, kernel, grid): duration, HBM bytes, instruction counts, fractions
   profiles/traffic.json              : per-launch HBM bytes and VALU wave-instructions bench.py reads (roofline.traffic, valu_frac),
                                        written FROM SCRATCH from this tag's passes (earlier rounds' figures stay in their own CSVs)
HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KB and, on gfx950, FETCH_SIZE tallies 16-byte-per-lane
streaming reads at half their size (MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
SRC = os.path.join(ROOT, "gpurun_out", f"{TAG}_pmc")
OUT = os.path.join(ROOT, "profiles")
VALU_PEAK = 256 * 4 * 32 * 2.4e9

# algorithmic bytes per launch of the profiled workloads (SURVEY 8(d) per-unit figures x units; tools/prof_roi.py sizes)
ALGO = {
    ("fwd", "k_roipool_fwd"): 512000 * 200704 + 256 * 3891200 + 512000 * 20,
    ("bwd", "k_roipool_bwd"): 512000 * 200704 + 256 * 3891200,
    ("fwdwide", "k_roipool_fwd"): 64000 * 200704 + 32 * 3891200 + 64000 * 20,
    ("kmeans", "k_kmeans_step"): 12 * 10_000_000,
    ("kmeans_prepared", "k_kc_iter"): 12 * 10_000_000,        # SURVEY's bytes as an equivalent rate: the kernel reads pieces, not boxes
    ("kmeans_shard", "k_kmeans_step"): 12 * 1_250_000,
    ("nms64k", "k_nms_mask"): 65536 * 65536 // 16 + 20 * 65536,
    ("nms256", "k_nms_mask"): 256 * (2048 * 2048 // 16 + 20 * 2048),
    ("pairwise", "k_iou_pairwise_f32"): 4 * 65536 * 65536 + 32 * 65536,
}
UNITS = {"fwd": 512000, "bwd": 512000, "fwdwide": 64000, "kmeans": 10_000_000, "kmeans_prepared": 10_000_000, "kmeans_shard": 1_250_000,
         "nms64k": 65536, "nms64k_dense": 65536, "nms256": 512000, "nms2000": 2000, "pairwise": 65536}

rows = collections.defaultdict(dict)          # (workload, kernel, grid) -> {counter: value, "us": duration}
per_workload = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(SRC, "*.csv"))):
    wl, _ = os.path.basename(f)[:-4].rsplit("_", 1)
    for r in csv.DictReader(open(f)):
        k = r["kernel"]
        if not k.startswith("k_"):
            continue
        per_workload[wl].append(r)
        d = rows[(wl, k, r["grid_threads"])]
        d[r["counter"]] = float(r["mean_value"])
        d.setdefault("us", []).append(float(r["avg_us"]))
        d["launches"] = r["launches"]

for wl, rs in per_workload.items():
    with open(os.path.join(OUT, f"{TAG}_pmc_{wl}.csv"), "w") as fo:
        fo.write("kernel,grid_threads,counter,launches,mean_value,avg_us\n")
        for r in rs:
            fo.write(f"\"{r['kernel']}\",{r['grid_threads']},{r['counter']},{r['launches']},{r['mean_value']},{r['avg_us']}\n")

cols = ["FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
        "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY",
        "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"]
traffic_path = os.path.join(OUT, "traffic.json")
traffic = {"_how": f"tools/profile_pmc.sh + tools/pmc_collect.py {TAG}: rocprofv3 --pmc, one pass per run (FETCH_SIZE; WRITE_SIZE; two SQ passes; GRBM), "
                   f"python3 tools/prof_roi.py <workload>; values per launch; HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, FETCH_SIZE doubled as "
                   f"MI355X_MICROARCH.md prescribes for gfx950 (16-byte-per-lane streaming reads are tallied at half their size); "
                   f"sources: profiles/{TAG}_pmc_*.csv"}
traffic.setdefault("valu_insts_per_launch", {})
traffic.setdefault(f"{TAG}_hbm_bytes_per_launch", {})
seen_valu, seen_hbm = {}, {}
with open(os.path.join(OUT, f"{TAG}_pmc_summary.csv"), "w") as fo:
    fo.write("workload,kernel,grid_threads,launches,avg_us,hbm_bytes,algorithmic_bytes,traffic_ratio,hbm_frac_of_8TBps,valu_lane_ops_frac,"
             "valu_per_unit,wait_any_share,wait_inst_share,active_share," + ",".join(cols) + "\n")
    for (wl, k, g), d in sorted(rows.items()):
        us = sum(d["us"]) / len(d["us"])
        hbm = (2 * d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0)) * 1024 if "FETCH_SIZE" in d and "WRITE_SIZE" in d else None
        base = k.split("<")[0]
        algo = ALGO.get((wl, base))
        valu = d.get("SQ_INSTS_VALU")
        wc = d.get("SQ_WAVE_CYCLES")
        def share(c):
            return f"{d[c] / wc:.3f}" if (wc and c in d) else ""
        fo.write(",".join(str(x) for x in [
            wl, f"\"{k}\"", g, d["launches"], f"{us:.2f}", f"{hbm:.0f}" if hbm else "", algo or "",
            f"{hbm / algo:.3f}" if (hbm and algo) else "", f"{algo / (us * 1e-6) / 8e12:.3f}" if algo else "",
            f"{valu * 64 / (us * 1e-6) / VALU_PEAK:.3f}" if valu else "", f"{valu * 64 / UNITS[wl]:.1f}" if valu else "",
            share("SQ_WAIT_ANY"), share("SQ_WAIT_INST_ANY"), share("SQ_ACTIVE_INST_ANY")] + [f"{d[c]:.6g}" if c in d else "" for c in cols]) + "\n")
        # instantiations of one kernel that together serve a launch of the operator (k_roipool_fwd<.., 0> and <.., 1>) are summed
        if us > 5 and valu:
            seen_valu[(wl, base)] = seen_valu.get((wl, base), 0.0) + valu
            traffic["valu_insts_per_launch"].setdefault(base, {})[str(UNITS[wl])] = seen_valu[(wl, base)]
        if hbm and algo:
            seen_hbm[(wl, base)] = seen_hbm.get((wl, base), 0.0) + hbm
            traffic[f"{TAG}_hbm_bytes_per_launch"].setdefault(base, {})[str(UNITS[wl])] = {"hbm_bytes": seen_hbm[(wl, base)], "algorithmic_bytes": algo, "workload": wl}
fwd = traffic[f"{TAG}_hbm_bytes_per_launch"].get("k_roipool_fwd", {}).get("512000")
if fwd:
    traffic.setdefault("k_roipool_fwd_bytes_per_launch", {})["512000"] = int(fwd["hbm_bytes"])
bwd = traffic[f"{TAG}_hbm_bytes_per_launch"].get("k_roipool_bwd", {}).get("512000")
if bwd:
    traffic.setdefault("k_roipool_bwd_bytes_per_launch", {})["512000"] = int(bwd["hbm_bytes"])
json.dump(traffic, open(traffic_path, "w"), indent=1)
print(open(os.path.join(OUT, f"{TAG}_pmc_summary.csv")).read())
