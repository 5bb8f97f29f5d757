"""Where a ONE-TILE step goes (small batches: cfg1, one rank's shard of an 8-GPU cfg2 plan): rollout time per mode and the
in-kernel phase profile of workgroup 0 (generic instance; the lean instances compile the profiler out).  Run on a GPU box
from the repo root; HIPETS_LIB selects a library build.  Prints one JSON line."""
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

PHASES = {0: "prologue", 7: "k-split share (-DHIPETS_LEAN_PROF builds of the fused one-tile instances)", 8: "layer barrier", 9: "sample", 10: "reward+next input", 11: "k loop", 12: "dispatch", 13: "epilogue", 14: "set-up"}
dev = torch.device("cuda:0")
eng = hipets.get_engine(dev)
out = {"lib": os.environ.get("HIPETS_LIB", "default")}


def kernel_ms(fn, n=20):
    for _ in range(5):
        fn(0)
    eng.timing_enable(True)
    eng.timing_read(reset=True)
    for i in range(n):
        fn(i + 1)
    cnt, ms = eng.timing_read(reset=True)
    eng.timing_enable(False)
    return ms / n, cnt // n


for name, obs, act, pop, H, P, rew in [("cfg1_cartpole", 4, 1, 100, 15, 5, "cartpole"), ("cfg2_shard_of_8", 17, 6, 63, 30, 20, "halfcheetah"),
                                       ("cfg2_shard_of_4", 17, 6, 125, 30, 20, "halfcheetah"), ("cfg2_shard_of_2", 17, 6, 250, 30, 20, "halfcheetah")]:
    # (cartpole with its termination function, as the reference's config has it: the model then matches its shape-specialised instance;
    # rounds 2 and 3 first timed it without, i.e. on the generic kernel)
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=200, seed=0, nontrivial_stats=False, reward=rew,
                                 termination="cartpole" if rew == "cartpole" else "no_termination")
    eng.set_model(to_spec(om, obs, act))
    acts = (torch.rand(pop, H, act) * 2 - 1).to(dev)
    s0 = np.zeros(obs, np.float32)
    nwg, r = eng.fast_geometry(pop, P, H, 0)
    rec = {"workgroups_fast": nwg, "row_tiles_per_workgroup": r}
    for mode in ("fast", "device"):
        ms, launches = kernel_ms(lambda i: eng.rollout(acts, s0, P, mode=mode, seed=1, stream_id=i))
        rec[mode] = {"rollout_kernel_ms": ms, "launches": launches, "us_per_step": 1e3 * ms / H}
    ms, _ = kernel_ms(lambda i: eng.rollout(acts, s0, P, mode="fast", seed=1, stream_id=i, generic_kernel=True))
    rec["fast_generic_kernel_ms"] = ms
    pc = torch.zeros(8, 16, dtype=torch.int64, device=dev)
    eng.rollout(acts, s0, P, mode="fast", seed=1, stream_id=99, phase_cycles=pc)
    torch.cuda.synchronize()
    pcs = pc.cpu()
    for w in (0, 3):
        rec[f"phase_cycles_per_step_wave{w}"] = {PHASES[k]: int(pcs[w, k]) // H for k in PHASES}
        rec[f"cycles_per_step_wave{w}"] = int(pcs[w].sum()) // H
    out[name] = rec
print(json.dumps(out))
