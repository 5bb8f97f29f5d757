"""The row-tile rule against the committed forced-R sweeps (profiles/r4_device_r_sweep.json, r4_stock_workloads.json): for every measured
workload and mode, the R the rule picks must be the fastest measured one or within 1 % of it."""
import json
import os

import cost_model_replica as cm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def measured(fname, key, mode):
    d = json.load(open(os.path.join(ROOT, "profiles", fname)))[key]
    if fname == "r4_device_r_sweep.json":
        return {R: d[f"{mode}_R{R}"]["ms"] for R in (1, 2, 3, 4) if "ms" in d[f"{mode}_R{R}"]}
    return {R: d[mode][f"R{R}"]["rollout_kernel_ms"] for R in (1, 2, 3, 4) if "rollout_kernel_ms" in d[mode].get(f"R{R}", {})}


def test_rule_picks_the_fastest_measured_row_tile_count():
    cases, optimal, worst = 0, 0, 0.0
    for (fname, key), (pop, P, members, lean) in cm.WORKLOADS.items():
        for mode in ("fast", "device"):
            ms = measured(fname, key, mode)
            assert len(ms) == 4, (key, mode)
            pick = cm.choose_r(pop, P, members, mode, lean)
            best = min(ms, key=ms.get)
            regret = ms[pick] / ms[best] - 1.0
            cases += 1
            optimal += pick == best
            worst = max(worst, regret)
            assert regret <= 0.01, (key, mode, pick, best, ms)
    assert cases == 24 and optimal >= 23 and worst <= 0.01


def test_the_mode_blind_rule_it_replaced_was_worse_on_the_same_data():
    """Without the desynchronised-pair term (FAST priced like DEVICE) the pop-1000 FAST batch gets three row tiles: > 10 % slower."""
    pop, P, members, lean = cm.WORKLOADS[("r4_device_r_sweep.json", "cfg2 x 2 (pop 1000 x 20, H 30)")]
    ms = measured("r4_device_r_sweep.json", "cfg2 x 2 (pop 1000 x 20, H 30)", "fast")
    tiles = (pop + 15) // 16
    blind = min((1, 2, 3, 4), key=lambda R: (cm.cost(tiles, P, R, R in lean, False), R))
    assert blind == 3 and cm.choose_r(pop, P, members, "fast", lean) == 1
    assert ms[3] / ms[1] > 1.10


def test_wide_instances_get_the_faster_of_their_two_row_tile_counts():
    """The Humanoid-v4 (WIDE) instances exist for one and two row tiles per workgroup: on the cfg4' iCEM plan's five population sizes, both
    modes, the rule's pick is the faster measured one (profiles/r5_cfg4p_iterations.json, the session with FAST rows dealt as one run).
    Priced like the narrow instances -- the rule of round 4 -- it picked the slower one three times on the same sizes."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r5_cfg4p_iterations.json")))
    after = d["after (session r5n: FAST rows dealt as one run, WIDE cost model)"]
    old_misses = 0
    for pop, rec in after.items():
        for mode in ("fast", "device"):
            ms = {R: rec[mode][f"R{R}"]["ms"] for R in (1, 2)}
            pick = cm.choose_r(int(pop), 20, 5, mode, {1, 2}, rs=(1, 2), wide=True)
            assert pick == rec[mode]["rule_picks"][1] == min(ms, key=ms.get), (pop, mode, pick, ms)
            fast = mode == "fast"
            tiles, slices = ((int(pop) * 20 + 15) // 16, 1) if fast else ((int(pop) * 4 + 15) // 16, 5)
            narrow = min((1, 2), key=lambda R: (cm.cost(tiles, slices, R, True, fast), R))  # (and two workgroups of R <= 2 per CU, which WIDE cannot)
            old_misses += narrow != pick
    assert old_misses >= 3


def test_rule_on_the_shipped_workloads_with_fast_rows_dealt_as_one_run():
    """Round 5's sweep of the shipped workloads (profiles/r5_stock_workloads.json: every forced R, FAST geometry as one run): the rule's
    pick is what the library picked in that session, it is the fastest measured R in all but at most two of the sixteen (workload, mode)
    cases, and where it is not (pets_hopper FAST is the known miss: R = 2 measures 2-3 % faster than the picked R = 1) it loses at most
    3 % -- plus the sweep's OWN run-to-run spread: every file holds the pick's configuration twice, as "default" and as the forced
    "R<pick>", and the two readings differ by up to a few tenths of a per cent.  (Round 5 asserted a bare 2.5 % against a file the
    closing evidence session re-measured afterwards: 2.58 % on that box, and the suite went red with nothing changed -- round-5 verdict.)"""
    d = json.load(open(os.path.join(ROOT, "profiles", "r5_stock_workloads.json")))
    lean = {"cfg2_synthetic": {1, 2, 3}, "stock_halfcheetah": {1, 2, 3}, "stock_inv_pendulum": {3}}
    pops = {"cfg2_synthetic": 500, "stock_halfcheetah": 400, "stock_inv_pendulum": 480}
    regrets, spread = {}, 0.0
    for key, rec in d.items():
        for mode in ("fast", "device"):
            ms = {R: rec[mode][f"R{R}"]["rollout_kernel_ms"] for R in (1, 2, 3, 4) if "rollout_kernel_ms" in rec[mode].get(f"R{R}", {})}
            pick = cm.choose_r(pops.get(key, 350), 20, 5, mode, lean.get(key, {1, 2}))
            assert pick == rec["kernel_class"][mode][1], (key, mode)
            regrets[(key, mode)] = ms[pick] / min(ms.values()) - 1.0
            spread = max(spread, abs(rec[mode]["default"]["rollout_kernel_ms"] / ms[pick] - 1.0))
    assert len(regrets) == 16 and spread < 0.01
    missed = {k: round(v, 4) for k, v in regrets.items() if v > spread}  # (a "miss" inside the run-to-run spread is a tie)
    assert len(missed) <= 2, missed
    assert max(regrets.values()) <= 0.03 + spread, missed
