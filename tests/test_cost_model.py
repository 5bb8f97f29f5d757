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
