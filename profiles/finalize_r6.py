"""Turn the outputs of round 6's GPU sessions (gpurun_out/r6*) into the committed summaries under profiles/ (run on the build box after
`profiles/session_r6_final.sh`; then `python -m pytest tests -q -m "not gpu"` -- profiles/finalize_r6.sh does both: the suite reads some of these files)."""
import csv
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def jload(path, start='{'):
    t = open(path).read()
    return json.loads(t[t.index(start):t.rindex('}') + 1])


def copy(src, dst):
    if os.path.exists(os.path.join(G, src)) and os.path.getsize(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, dst))
        return True
    print("missing", src)
    return False


F = "r6_final"
copy(f"{F}/tests.log", "r6_gpu_tests.log")
copy(f"{F}/bench_line.json", "r6_bench_line.json")
copy(f"{F}/bench_line_gloo2.json", "r6_bench_line_gloo2.json")
copy(f"{F}/one_tile_phase_profile.json", "r6_one_tile_phase_profile.json")
copy(f"{F}/small_batches.json", "r6_small_batches.json") or copy(f"{F}/small_batches.log", "r6_small_batches.log")
# turn trace: the shipped collect (LDS-DMA, rows dealt to the waves) next to the round's earlier versions
FIRST = "first closing session of round 6 (commit 2624dd1: before the ragged last turn and the paired draws)"


def carried(name, old_key):
    """the entry the FIRST closing session of the round left in the committed summary (that session's gpurun_out files are overwritten)"""
    try:
        old = json.load(open(os.path.join(P, name)))
        return old.get(FIRST) or old.get(old_key)
    except Exception as exc:
        print("carried", name, exc)
        return None


tt = {"what": "profiles/turn_trace.py on -DHIPETS_STEP_TRACE builds: where a (step, turn) of the turn-based persistent DEVICE form goes, us"}
tt_first = carried("r6_turn_trace.json", "shipped (LDS-DMA collect by row slot, input pass by column; closing session)")
for tag, path in (("shipped (closing session: + ragged last turn, paired draws)", f"{F}/turn_trace.log"),
                  ("shipped, cfg4' pop 497 / pop 1001: a ragged last turn (closing session)", f"{F}/turn_trace_ragged.log"),
                  ("r6g: the same, session r6g", "r6g/turn_trace_dma.log"),
                  ("r6f: LDS-DMA collect by row slot, first cut; input pass by column", "r6f/turn_trace_dma.log"),
                  ("r6e: LDS-DMA collect by row slot; input pass by group (round 4-5)", "r6e/turn_trace_dma.log"),
                  ("r6d: LDS-DMA collect, FIRST version (chunks of 64 consecutive items per wave)", "r6d/turn_trace_dma.log"),
                  ("r6d: register path of rounds 3-5 (-DHIPETS_DMA_COLLECT=0), same box as the line above", "r6d/turn_trace_nodma.log")):
    try:
        tt[tag] = {k: v for k, v in jload(os.path.join(G, path)).items() if isinstance(v, dict)}
    except Exception as exc:
        print("turn trace", path, exc)
if tt_first:
    tt[FIRST] = tt_first
json.dump(tt, open(os.path.join(P, "r6_turn_trace.json"), "w"), indent=1)
# cfg4' per population size: shipped vs the A/B builds of the sessions that introduced each change
ci = {"what": "profiles/cfg4p_iteration_probe.py: the five population sizes of the cfg4' iCEM plan as single rollouts (ms, fraction of the fp32 peak)"}
ci_first = carried("r6_cfg4p_iterations.json", "shipped (closing session)")
for tag, path in (("shipped (closing session: + ragged last turn, paired draws)", f"{F}/cfg4p_iterations.json"), ("r6g", "r6g/cfg4p_iterations.json"),
                  ("r6f: shipped collect, input pass by column", "r6f/cfg4p_iterations.json"),
                  ("r6f: -DHIPETS_INPUT_BY_COLUMN=0, same box", "r6f/cfg4p_iterations_nocols.json"),
                  ("r6c: LDS-DMA collect FIRST version", "r6c/cfg4p_iterations.json"),
                  ("r6c: -DHIPETS_DMA_COLLECT=0 (the round-5 register path), same box", "r6c/cfg4p_iterations_nodma.json")):
    try:
        d = json.load(open(os.path.join(G, path)))
        ci[tag] = {pop: {m: d[pop][f"{m}_R0"] for m in ("device", "fast")} for pop in ("1036", "805", "630", "497", "358")}
        ci[tag]["sum_ms"] = {m: round(sum(d[pop][f"{m}_R0"]["ms"] for pop in ("1036", "805", "630", "497", "358")), 3) for m in ("device", "fast")}
    except Exception as exc:
        print("cfg4p", path, exc)
if ci_first:
    ci[FIRST] = ci_first
json.dump(ci, open(os.path.join(P, "r6_cfg4p_iterations.json"), "w"), indent=1)
# headline bound: shipped vs the timing-only no-draws build (session r6d), and the closing session's five probe runs
hb = {"what": "profiles/headline_probe.py: cfg2 rollout kernel, ms per launch (median of 5 blocks of 20 launches per run)"}
hb_first = carried("r6_headline_bound.json", "shipped library (closing session)")
for tag, pat, n in (("shipped library (session r6d)", "r6d/headline_%d.log", 3),
                    ("timing-only build, tails draw nothing: -DHIPETS_TIMING_NO_DRAWS=1 (session r6d, same box)", "r6d/headline_nodraws_%d.log", 3),
                    ("shipped library (closing session: + paired draws)", f"{F}/headline_%d.log", 5)):
    dev, fast = [], []
    for i in range(1, n + 1):
        try:
            d = jload(os.path.join(G, pat % i), '{"lib"')
            dev.append(d["device"]["median_ms"])
            fast.append(d["fast"]["median_ms"])
        except Exception as exc:
            print("headline", pat % i, exc)
    if dev:
        hb[tag] = {"device_ms": dev, "fast_ms": fast, "device_median": statistics.median(dev), "fast_median": statistics.median(fast),
                   "device_minus_fast_pct": 100 * (statistics.median(dev) / statistics.median(fast) - 1)}
if hb_first:
    hb[FIRST] = hb_first
json.dump(hb, open(os.path.join(P, "r6_headline_bound.json"), "w"), indent=1)
# small plans: kernels per plan against wall time per plan (session r6b)
sp = {"what": "rocprofv3 --kernel-trace --stats of profiles/other_configs.py --only <plan> (back-to-back plans) and profiles/plan_gap_probe.py: is the GPU idle "
              "between the kernels of a small plan?"}
for tag, d in (("cfg1_cem_plan_device", "r6b/trace_cfg1_cem_plan_device"), ("cfg1_cem_plan_fast", "r6b/trace_cfg1_cem_plan_fast"), ("planet_cem_plan", "r6b/trace_planet_device")):
    try:
        rows = list(csv.DictReader(open(os.path.join(G, d, "t_kernel_stats.csv"))))
        sp[tag] = [{"kernel": r["Name"][:90], "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2), "pct": float(r["Percentage"])} for r in rows[:8]]
    except Exception as exc:
        print("small plans", d, exc)
try:
    sp["plan_gap_probe"] = {k: v for k, v in jload(os.path.join(G, "r6b/plan_gaps.log")).items() if k != "env"}
except Exception as exc:
    print("plan gaps", exc)
json.dump(sp, open(os.path.join(P, "r6_small_plan_kernel_stats.json"), "w"), indent=1)
try:
    t = open(os.path.join(G, f"{F}/pair_exchange.log")).read()
    json.dump(json.loads(t[t.index('{'):t.index('}}') + 2]), open(os.path.join(P, "r6_pair_exchange.json"), "w"), indent=1)
except Exception as exc:
    print("pair exchange", exc)
    copy("r6e/pair_exchange.log", "r6_pair_exchange.log")
print("done; now: python profiles/summarize.py r6")
