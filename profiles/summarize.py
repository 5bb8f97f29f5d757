#!/usr/bin/env python
"""Turn the rocprofv3 CSVs of profiles/collect.sh (gpurun_out/prof_<tag>/) into the committed, judged summaries:

  profiles/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary of `python bench.py` (verbatim)
  profiles/<tag>_rollout_pmc.json      per-launch PMC averages of hipets::rollout_kernel + derived figures
  profiles/hbm_traffic.json            HBM bytes per rollout_kernel launch, read by bench.py (roofline.traffic)

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE come from separate
--pmc passes, are in KiB, and on gfx950 FETCH_SIZE reports exactly half the bytes of wide (16 B/lane) coalesced
streaming reads -- which is how this kernel reads its weights -- so the read side is doubled.
"""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")

shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_line.json"), os.path.join(dst, f"{tag}_bench_line_under_rocprof.json"))

counters = defaultdict(list)
meta = {}
for sub in ("pmc_sq", "pmc_lds", "pmc_fetch", "pmc_write"):
    path = os.path.join(src, sub, "pmc_counter_collection.csv")
    if not os.path.exists(path):
        continue
    for row in csv.DictReader(open(path)):
        if "rollout_kernel" not in row["Kernel_Name"]:
            continue
        counters[row["Counter_Name"]].append(float(row["Counter_Value"]))
        meta = {k: row[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                     "Accum_VGPR_Count", "SGPR_Count")}
avg = {k: sum(v) / len(v) for k, v in counters.items()}
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(src, "trace", "bench_kernel_stats.csv")))}
roll = next(v for k, v in stats.items() if "rollout_kernel" in k)
avg_ns = float(roll["AverageNs"])
out = {"tag": tag, "kernel": meta, "launches_profiled": {k: len(v) for k, v in counters.items()}, "per_launch_avg": avg,
       "kernel_trace": {"calls": int(roll["Calls"]), "avg_ns": avg_ns, "min_ns": float(roll["MinNs"]), "max_ns": float(roll["MaxNs"]),
                        "pct_of_gpu_time": float(roll["Percentage"])}}
d = {}
if "SQ_WAVE_CYCLES" in avg:
    wc = avg["SQ_WAVE_CYCLES"]  # quad-cycles summed over waves
    d["wait_any_frac"] = avg["SQ_WAIT_ANY"] / wc
    d["wait_inst_any_frac (MFMA pipe / issue stalls)"] = avg["SQ_WAIT_INST_ANY"] / wc
    d["active_inst_any_frac"] = avg["SQ_ACTIVE_INST_ANY"] / wc
    d["mfma_instructions"] = avg["SQ_INSTS_MFMA"]
    d["mfma_busy_cycles_32_per_inst"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"]
if "GRBM_GUI_ACTIVE" in avg:
    d["effective_clock_GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration)"] = avg["GRBM_GUI_ACTIVE"] / 8 / avg_ns
    waves = int(meta["Grid_Size"]) // 64
    d["mfma_pipe_busy_frac_on_active_simds"] = avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / waves / (avg["GRBM_GUI_ACTIVE"] / 8)
if "SQ_LDS_IDX_ACTIVE" in avg:
    d["lds_bank_conflict_frac"] = avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"]
traffic = None
if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
    fetch_b = avg["FETCH_SIZE"] * 1024 * 2  # KiB -> B, x2 gfx950 wide-read correction
    write_b = avg["WRITE_SIZE"] * 1024
    traffic = fetch_b + write_b
    d["hbm_read_bytes_per_launch (2 x FETCH_SIZE KiB)"] = fetch_b
    d["hbm_write_bytes_per_launch"] = write_b
    d["hbm_bytes_per_launch"] = traffic
out["derived"] = d
json.dump(out, open(os.path.join(dst, f"{tag}_rollout_pmc.json"), "w"), indent=1)
if traffic is not None:
    json.dump({"rollout_kernel_bytes_per_launch": traffic, "source": f"profiles/{tag}_rollout_pmc.json",
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KiB; read side x2 (gfx950)"},
              open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
print(json.dumps(out["kernel_trace"]))
