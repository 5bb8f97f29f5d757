"""Phase profile of the SHIPPED one-tile instances (round-5 verdict, item 3): the fused k-split instances that cfg1 and every rank of a
strong-scaled cfg2 plan run, and the PlaNet STATIC instance -- not the generic kernel's (whose profile rounds 2-5 committed under the
one-tile heading: the shape-specialised instances compile the profiler out unless built with -DHIPETS_LEAN_PROF=1).

    python profiles/build_variant.py leanprof rollout_r1.hip rollout_r1_fast.hip planet.hip -DHIPETS_LEAN_PROF=1     (build box)
    HIPETS_LIB=profiles/variants/leanprof.so python profiles/one_tile_phase_profile.py > gpurun_out/r6_one_tile_phase_profile.json

Per workload and mode: (1) the UNPROFILED launch duration of this very build (no phase buffer passed: the marks are dead branches),
(2) one launch with the phase buffer: per phase and wave the cycles and the NUMBER of marks (rollout.hpp Prof: count in the upper
bits), (3) the cost of a mark, calibrated as (profiled - unprofiled duration) / marks of wave 0, subtracted per phase.  Workgroup 0 is
the only one that stamps; in the profiled launch it is the slowest workgroup, so the launch duration is its timeline."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mbrl-lib_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import hipets  # noqa: E402
from conftest import to_spec  # noqa: E402
from oracle import pets_oracle as po  # noqa: E402  (random weights only)

sys.path.insert(0, ROOT)
import bench  # noqa: E402  (synthetic_planet_spec)

PHASES = {0: "prologue", 7: "k-split share", 8: "layer barrier", 9: "sample / GRU", 10: "reward + next input", 11: "k loop", 12: "dispatch",
          13: "epilogue", 14: "set-up"}
SHIFT = 44
dev = torch.device("cuda:0")
eng = hipets.get_engine(dev)
out = {"lib": os.environ.get("HIPETS_LIB", "default (the shipped library: its shape-specialised instances do not stamp)"),
       "what": __doc__.split("\n\n")[0]}


def event_ms(fn, n=20, warm=5):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(warm + i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def kernel_ms(fn, n=20, warm=5):
    """average rollout-KERNEL duration (the engine's own events on the dispatch packets: no launch gaps, no particle-mean kernel)"""
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    eng.timing_enable(True)
    eng.timing_read(reset=True)
    for i in range(n):
        fn(warm + i)
    torch.cuda.synchronize()
    cnt, ms = eng.timing_read(reset=True)
    eng.timing_enable(False)
    return ms / max(cnt, 1) * (cnt / n)  # (per rollout: one persistent launch, or H per-step launches)


def decode(pc, H, unprofiled_ms, profiled_ms):
    pcs = pc.cpu().numpy().astype(np.int64)
    cyc, cnt = pcs & ((1 << SHIFT) - 1), pcs >> SHIFT
    rec = {}
    tot0, marks0 = int(cyc[0, :15].sum()), int(cnt[0, :15].sum())
    # cost of a mark: what the stamping added to workgroup 0's timeline, per mark of its wave 0 (cycles of THIS launch's clock)
    mark = tot0 * (1.0 - unprofiled_ms / profiled_ms) / max(marks0, 1)
    rec["cycles_per_mark_calibrated"] = mark
    rec["clock_ghz_implied"] = tot0 / (profiled_ms * 1e6)
    for w in (0, 3):
        raw = {PHASES[k]: int(cyc[w, k]) / H for k in PHASES}
        n = {PHASES[k]: int(cnt[w, k]) / H for k in PHASES}
        rec[f"wave{w}"] = {"cycles_per_step_raw": raw, "marks_per_step": n,
                           "cycles_per_step_corrected": {k: raw[k] - n[k] * mark for k in raw},
                           "total_raw": sum(raw.values()), "total_corrected": sum(raw[k] - n[k] * mark for k in raw)}
    return rec


CASES = [("cfg1_cartpole (BASELINE configs[0]: pop 100 x 5, H 15)", 4, 1, 100, 15, 5, "cartpole"),
         ("cfg2_shard_of_8 (63 candidates x 20, H 30)", 17, 6, 63, 30, 20, "halfcheetah")]
# PHASE_CASES=cfg2: the HEADLINE workload instead (BASELINE configs[1]: pop 500 x 20, H 30 -- the R = 3 instances; a variant built with
#   python profiles/build_variant.py leanprof3 rollout_r3.hip rollout_r3_fast.hip -DHIPETS_LEAN_PROF=1), no PlaNet
HEADLINE = os.environ.get("PHASE_CASES") == "cfg2"
if HEADLINE:
    CASES = [("cfg2 (BASELINE configs[1]: pop 500 x 20, H 30; 209 three-tile workgroups)", 17, 6, 500, 30, 20, "halfcheetah")]
for name, obs, act, pop, H, P, rew in CASES:
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=200, seed=0, nontrivial_stats=False, reward=rew,
                                 termination="cartpole" if rew == "cartpole" else "no_termination")
    eng.set_model(to_spec(om, obs, act))
    acts = (torch.rand(pop, H, act) * 2 - 1).to(dev)
    s0 = np.zeros(obs, np.float32)
    rec = {}
    for mode in ("device", "fast"):
        cls = list(eng.kernel_class(pop, P, H, mode))
        ums = kernel_ms(lambda i: eng.rollout(acts, s0, P, mode=mode, seed=1, stream_id=i))
        pc = torch.zeros(8, 16, dtype=torch.int64, device=dev)
        pms = kernel_ms(lambda i: eng.rollout(acts, s0, P, mode=mode, seed=1, stream_id=100 + i, phase_cycles=pc), n=10, warm=2)
        pc.zero_()
        eng.rollout(acts, s0, P, mode=mode, seed=1, stream_id=999, phase_cycles=pc)
        torch.cuda.synchronize()
        stamped = bool(pc.any())
        rec[mode] = {"kernel_class": cls, "unprofiled_rollout_ms": ums, "us_per_step_unprofiled": 1e3 * ums / H, "profiled_rollout_ms": pms,
                     "instance_stamped": stamped}
        if stamped:
            rec[mode].update(decode(pc, H, ums, pms))
    out[name] = rec

if HEADLINE:
    print(json.dumps(out))
    sys.exit(0)
# PlaNet, conf/dynamics_model/planet.yaml sizes, pop 1000 x H 12 (conf/overrides/planet_cheetah_run.yaml): 63 one-tile workgroups
P_POP, P_H = 1000, 12
pspec = bench.synthetic_planet_spec(dev)
eng.planet_set_model(pspec)
acts = (torch.rand(P_POP, P_H, pspec.action_size) * 2 - 1).to(dev)
lat0, bel0 = torch.zeros(pspec.latent_size, device=dev), torch.zeros(pspec.belief_size, device=dev)
try:
    ums = event_ms(lambda i: eng.planet_rollout(acts, lat0, bel0, 1, seed=1, stream_id=i))
    pc = torch.zeros(8, 16, dtype=torch.int64, device=dev)
    pms = event_ms(lambda i: eng.planet_rollout(acts, lat0, bel0, 1, seed=1, stream_id=100 + i, phase_cycles=pc), n=10, warm=2)
    pc.zero_()
    eng.planet_rollout(acts, lat0, bel0, 1, seed=1, stream_id=999, phase_cycles=pc)
    torch.cuda.synchronize()
    rec = {"unprofiled_rollout_ms": ums, "us_per_step_unprofiled": 1e3 * ums / P_H, "profiled_rollout_ms": pms, "instance_stamped": bool(pc.any()),
           "note": "durations include the particle-mean launch behind the rollout kernel (~4 us)"}
    if rec["instance_stamped"]:
        rec.update(decode(pc, P_H, ums, pms))
except Exception as exc:  # (keep what the PETS workloads measured)
    rec = {"error": str(exc)[:300]}
out["planet_static (pop 1000 x H 12, one particle)"] = rec
print(json.dumps(out))
