#!/usr/bin/env python
"""Turn the rocprofv3 CSVs of profiles/collect.sh (gpurun_out/prof_<tag>/) into the committed, judged summaries:

  profiles/<tag>_kernel_stats_<mode>.csv   rocprofv3 --kernel-trace --stats summary of `python bench.py --mode <mode>` (verbatim)
  profiles/<tag>_bench_line_under_rocprof_<mode>.json   the bench line that profiled run printed
  profiles/<tag>_rollout_pmc.json          per-launch PMC averages of hipets::rollout_kernel + derived figures, per mode
  profiles/<tag>_config_kernel_stats.json  rollout / PlaNet kernel statistics of every other BASELINE configuration
  profiles/<tag>_other_configs.json        profiles/other_configs.py output (rollout and plan times, both modes)
  profiles/hbm_traffic.json                HBM bytes per rollout_kernel launch per mode, read by bench.py (roofline.traffic)

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE come from separate
--pmc passes, are in KiB, and on gfx950 FETCH_SIZE reports exactly half the bytes of wide (16 B/lane) coalesced
streaming reads -- which is how this kernel reads its weights -- so the read side is doubled.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")


def find(pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern), recursive=True))
    return hits[0] if hits else None


def rollout_row(stats_csv, names=("rollout_kernel",)):
    rows = [r for r in csv.DictReader(open(stats_csv)) if any(n in r["Name"] for n in names)]
    if not rows:
        return None
    calls = sum(int(r["Calls"]) for r in rows)
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    out = {"kernels": [r["Name"][:120] for r in rows], "calls": calls, "avg_ns": total / calls, "min_ns": min(float(r["MinNs"]) for r in rows),
           "max_ns": max(float(r["MaxNs"]) for r in rows), "pct_of_gpu_time": sum(float(r["Percentage"]) for r in rows)}
    # The persistent DEVICE form validates co-residency ONCE per process with a self-test launch of the very same kernel (a few
    # microseconds: every workgroup arrives at a counter and returns; rollout.hpp RolloutArgs::census).  rocprofv3 files it under
    # the rollout kernel's name; bench.py's hipEvents do not cover it.  Reported separately, so that the two averages compare.
    if len(rows) == 1 and out["min_ns"] < 0.05 * out["avg_ns"] and calls > 2:
        out["self_test_launches_excluded"] = 1
        out["self_test_ns"] = out["min_ns"]
        out["avg_ns_including_the_self_test_launch"] = out["avg_ns"]
        out["avg_ns"] = (total - out["min_ns"]) / (calls - 1)
        out["calls"] = calls - 1
    return out


out, traffic = {"tag": tag}, {}
for mode in ("device", "fast"):
    stats = find(f"trace_{mode}/**/bench_kernel_stats.csv")
    if not stats:
        continue
    shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats_{mode}.csv"))
    line = os.path.join(src, f"bench_line_{mode}.json")
    if os.path.exists(line) and os.path.getsize(line):
        shutil.copy(line, os.path.join(dst, f"{tag}_bench_line_under_rocprof_{mode}.json"))
    counters, meta = defaultdict(list), {}
    for sub in ("pmc_sq", "pmc_lds", "pmc_fetch", "pmc_write", "pmc_icache"):
        path = find(f"{sub}_{mode}/**/pmc_counter_collection.csv")
        if not path:
            continue
        for row in csv.DictReader(open(path)):
            if "rollout_kernel" not in row["Kernel_Name"]:
                continue
            counters[row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta = {k: row[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                         "Accum_VGPR_Count", "SGPR_Count") if k in row}
    # (the one co-residency self-test launch of DEVICE mode -- same kernel name, ~1e-3 of a rollout's counters -- is not a rollout)
    for k, v in list(counters.items()):
        med = sorted(v)[len(v) // 2]
        counters[k] = [x for x in v if not (med > 0 and x < 0.05 * med)]
    avg = {k: sum(v) / len(v) for k, v in counters.items()}
    roll = rollout_row(stats)
    d = {}
    if "SQ_WAVE_CYCLES" in avg:
        wc = avg["SQ_WAVE_CYCLES"]  # quad-cycles summed over waves
        d["wait_any_frac"] = avg["SQ_WAIT_ANY"] / wc
        d["wait_inst_any_frac (MFMA pipe / issue stalls)"] = avg["SQ_WAIT_INST_ANY"] / wc
        d["active_inst_any_frac"] = avg["SQ_ACTIVE_INST_ANY"] / wc
        d["mfma_instructions"] = avg["SQ_INSTS_MFMA"]
        d["valu_instructions_incl_mfma"] = avg["SQ_INSTS_VALU"]
        d["mfma_busy_cycles_32_per_inst"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"]
    if "GRBM_GUI_ACTIVE" in avg and roll and meta:
        d["effective_clock_GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration)"] = avg["GRBM_GUI_ACTIVE"] / 8 / roll["avg_ns"]
        waves = int(meta["Grid_Size"]) // 64
        d["mfma_pipe_busy_frac_on_active_simds"] = avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / waves / (avg["GRBM_GUI_ACTIVE"] / 8)
    if "SQ_LDS_IDX_ACTIVE" in avg:
        d["lds_bank_conflict_frac"] = avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"]
    if "SQC_ICACHE_REQ" in avg and avg["SQC_ICACHE_REQ"]:
        d["icache_miss_frac"] = avg.get("SQC_ICACHE_MISSES", 0.0) / avg["SQC_ICACHE_REQ"]
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        fetch_b = avg["FETCH_SIZE"] * 1024 * 2  # KiB -> B, x2 gfx950 wide-read correction
        write_b = avg["WRITE_SIZE"] * 1024
        traffic[f"rollout_kernel_bytes_per_launch_{mode}"] = fetch_b + write_b
        d["hbm_read_bytes_per_launch (2 x FETCH_SIZE KiB)"] = fetch_b
        d["hbm_write_bytes_per_launch"] = write_b
        d["hbm_bytes_per_launch"] = fetch_b + write_b
    out[mode] = {"kernel": meta, "launches_profiled": {k: len(v) for k, v in counters.items()}, "per_launch_avg": avg, "kernel_trace": roll,
                 "derived": d}
json.dump(out, open(os.path.join(dst, f"{tag}_rollout_pmc.json"), "w"), indent=1)
if traffic:
    traffic.update(source=f"profiles/{tag}_rollout_pmc.json",
                   method="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KiB; read side x2 (gfx950)")
    json.dump(traffic, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)

cfgs = {}
for d_ in sorted(glob.glob(os.path.join(src, "cfg_*"))):
    if not os.path.isdir(d_):
        continue
    stats = sorted(glob.glob(os.path.join(d_, "**", "t_kernel_stats.csv"), recursive=True))
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        cfgs[os.path.basename(d_)[4:]] = {"dominant": rollout_row(stats[0], ("rollout_kernel", "planet_rollout_kernel")),
                                          "top_kernels": [{"name": r["Name"][:100], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                                           "pct": float(r["Percentage"])} for r in rows[:5]]}
if cfgs:
    json.dump(cfgs, open(os.path.join(dst, f"{tag}_config_kernel_stats.json"), "w"), indent=1)
sb = os.path.join(src, "small_batches.json")
if os.path.exists(sb) and os.path.getsize(sb):
    shutil.copy(sb, os.path.join(dst, f"{tag}_small_batches.json"))
tcc = sorted(glob.glob(os.path.join(src, "pmc_tcc_bf16x3_device", "**", "pmc_counter_collection.csv"), recursive=True))
if tcc:
    acc = {}
    for r in csv.DictReader(open(tcc[0])):
        if "rollout_kernel" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    per = {k: sum(v[-3:]) / len(v[-3:]) for k, v in acc.items()}
    if per.get("TCC_REQ_sum"):
        per["l2_hit_frac"] = per.get("TCC_HIT_sum", 0.0) / (per.get("TCC_HIT_sum", 0.0) + per.get("TCC_MISS_sum", 1.0))
    json.dump({"workload": "cfg2 rollout, precision bf16x3, DEVICE mode (persistent form, XCD-major workgroup order)", "per_launch": per},
              open(os.path.join(dst, f"{tag}_bf16x3_l2.json"), "w"), indent=1)
for w in ("stock_halfcheetah", "stock_cartpole", "stock_pusher"):  # the raw rocprofv3 statistics of the shipped workloads' rollouts, verbatim
    for mode in ("device", "fast"):
        st = sorted(glob.glob(os.path.join(src, f"cfg_{w}_{mode}", "**", "t_kernel_stats.csv"), recursive=True))
        if st:
            shutil.copy(st[0], os.path.join(dst, f"{tag}_{w}_kernel_stats_{mode}.csv"))
sw = os.path.join(src, "stock_workloads.json")
if os.path.exists(sw) and os.path.getsize(sw):
    shutil.copy(sw, os.path.join(dst, f"{tag}_stock_workloads.json"))
oc = os.path.join(src, "other_configs.json")
if os.path.exists(oc) and os.path.getsize(oc):
    shutil.copy(oc, os.path.join(dst, f"{tag}_other_configs.json"))
print(json.dumps({m: out[m]["derived"] for m in ("device", "fast") if m in out}, indent=1))
print(json.dumps({m: out[m]["kernel_trace"] for m in ("device", "fast") if m in out}))
