"""Kernel-variant measurements on the GPU box (run from the repo root; HIPETS_LIB selects the library build):
cfg2 rollout time per row-tile count R in FAST mode (hipEvents on the dispatch packets), the DEVICE-mode per-step launch,
and the in-kernel phase profile of workgroup 0.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mbrl-lib_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import hipets  # noqa: E402

PHASES = {0: "prologue", 8: "layer barrier", 9: "sample", 10: "reward+next input", 11: "k loop", 12: "dispatch", 13: "epilogue", 14: "set-up"}


def main():
    dev = torch.device("cuda:0")
    eng = hipets.get_engine(dev)
    spec = bench.synthetic_spec(dev)
    eng.set_model(spec)
    pop, H, P = bench.POP, bench.HORIZON, bench.PARTICLES
    g = torch.Generator().manual_seed(0)
    actions = (torch.rand(pop, H, bench.ACT, generator=g) * 2 - 1).to(dev)
    s0 = np.zeros(bench.OBS, np.float32)
    flops = spec.flops_per_candidate_step() * pop * P * H
    out = {"lib": os.environ.get("HIPETS_LIB", "default"), "fast": {}, "device": {}}
    for R in (1, 2, 3, 4):
        try:
            for _ in range(3):
                eng.rollout(actions, s0, P, mode="fast", seed=1, stream_id=1, rows_per_group=R)
            eng.timing_enable(True)
            eng.timing_read(reset=True)
            for i in range(10):
                eng.rollout(actions, s0, P, mode="fast", seed=1, stream_id=2 + i, rows_per_group=R)
            n, ms = eng.timing_read(reset=True)
            eng.timing_enable(False)
            nwg, _ = eng.fast_geometry(pop, P, H, R)
            out["fast"][f"R{R}"] = {"ms": ms / n, "workgroups": nwg, "frac_fp32_peak": flops / (ms / n * 1e-3) / 1e12 / bench.PEAK_FP32_TFLOPS}
        except Exception as exc:  # a variant that does not fit is a result too
            out["fast"][f"R{R}"] = {"error": str(exc)[:120]}
    for R in (0, 2, 3):
        for _ in range(2):
            eng.rollout(actions, s0, P, mode="device", seed=1, stream_id=1, rows_per_group=R)
        eng.timing_enable(True)
        eng.timing_read(reset=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(5):
            eng.rollout(actions, s0, P, mode="device", seed=1, stream_id=2 + i, rows_per_group=R)
        e1.record()
        torch.cuda.synchronize()
        n, ms = eng.timing_read(reset=True)
        eng.timing_enable(False)
        torch.cuda.synchronize()
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record()
        for i in range(5):
            eng.rollout(actions, s0, P, mode="device", seed=1, stream_id=2 + i, rows_per_group=R)
        f1.record()
        torch.cuda.synchronize()
        out["device"][f"R{R}"] = {"kernel_us_per_step": 1e3 * ms / n, "launches": n, "rollout_ms_kernels": ms / 5, "rollout_ms_wall_events_on_every_packet": e0.elapsed_time(e1) / 5,
                                  "rollout_ms_wall": f0.elapsed_time(f1) / 5}
    for _ in range(3):
        eng.rollout(actions, s0, P, mode="fast", seed=1, stream_id=1, generic_kernel=True)
    eng.timing_enable(True)
    eng.timing_read(reset=True)
    for i in range(10):
        eng.rollout(actions, s0, P, mode="fast", seed=1, stream_id=2 + i, generic_kernel=True)
    n, ms = eng.timing_read(reset=True)
    eng.timing_enable(False)
    out["fast"]["R3_generic_kernel"] = {"ms": ms / n}
    pc = torch.zeros(8, 16, dtype=torch.int64, device=dev)
    eng.rollout(actions, s0, P, mode="fast", seed=1, stream_id=99, phase_cycles=pc)
    torch.cuda.synchronize()
    pcs = pc.cpu()
    out["phase_cycles_per_step_wave0"] = {PHASES[k]: int(pcs[0, k]) // H for k in PHASES}
    out["phase_cycles_per_step_wave3"] = {PHASES[k]: int(pcs[3, k]) // H for k in PHASES}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
