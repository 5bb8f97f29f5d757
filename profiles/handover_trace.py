"""Where the persistent DEVICE form's step-to-step hand-over time goes (cfg2).  Needs the profiling build of the library
(python -c "import __graft_entry__ as g, os; g.build_library(out=os.path.join(g.PKG,'build_trace','libhipets_trace.so'),
extra_flags=['-DHIPETS_STEP_TRACE'], objdir=os.path.join(g.PKG,'build_trace'))") selected with HIPETS_LIB.  Every workgroup
stamps the chip-wide 100 MHz clock at four points of every step: MLP done, published, rows arrived, next input built.
Prints one JSON line (microseconds)."""
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

dev = torch.device("cuda:0")
eng = hipets.get_engine(dev)
eng.set_model(bench.synthetic_spec(dev))
g = torch.Generator().manual_seed(0)
pop, H, P = int(os.environ.get("HANDOVER_POP", bench.POP)), bench.HORIZON, bench.PARTICLES
actions = (torch.rand(pop, H, bench.ACT, generator=g) * 2 - 1).to(dev)
s0 = np.zeros(bench.OBS, np.float32)
NWG = 256
buf = torch.zeros(128 + NWG * H * 4 * 4, dtype=torch.int64, device=dev)  # the kernel's record stride: H * 4 records per workgroup
for i in range(200):  # warm the clocks up (the first rollouts after idle run ~5 % slower)
    eng.rollout(actions, s0, P, mode="device", seed=1, stream_id=i)
res = []
for rep in range(5):
    buf.zero_()
    eng.rollout(actions, s0, P, mode="device", seed=1, stream_id=10 + rep, phase_cycles=buf[:128].view(8, 16))
    torch.cuda.synchronize()
    st = buf[128:].view(NWG, H * 4, 4).cpu().numpy().astype(np.float64)[:, :H] * 0.01  # 100 MHz ticks -> us (straight form: one record per step)
    live = st[:, 0, 0] > 0
    st = st[live]
    n = st.shape[0]
    t0 = st[:, 0, 0].min()
    mlp_done, published, arrived, built = (st[:, :H - 1, k] for k in range(4))
    step_len = np.diff(st[:, :, 0], axis=1)  # MLP-done to MLP-done, per workgroup
    res.append({
        "workgroups": int(n),
        "rollout_us (first MLP done -> last MLP done + tail)": float(st[:, H - 1, 0].max() - t0),
        "step_us mean over workgroups and steps": float(step_len.mean()),
        "tail phases (MLP done -> published) mean": float((published - mlp_done).mean()),
        "wait for rows (published -> arrived) mean / p50 / p90 / max": [float(x) for x in ((arrived - published).mean(), np.percentile(arrived - published, 50), np.percentile(arrived - published, 90), (arrived - published).max())],
        "input build (arrived -> built) mean": float((built - arrived).mean()),
        "skew of 'published' across workgroups within a step: p90-p10 / max-min, mean over steps": [float((np.percentile(published, 90, axis=0) - np.percentile(published, 10, axis=0)).mean()), float((published.max(0) - published.min(0)).mean())],
        "last publisher -> last arrival, mean over steps (pure hand-over latency after the slowest producer)": float((arrived.max(0) - published.max(0)).mean()),
        "earliest arrival - latest publish (negative: somebody already had its rows before the slowest finished)": float((arrived.min(0) - published.max(0)).mean()),
        "MLP + tail of a step (built(t-1) -> published(t)) mean": float((published[:, 1:] - built[:, :-1]).mean()),
        "per-XCD mean step length": [float(step_len[np.arange(n) % 8 == x].mean()) for x in range(8)],
        # is a workgroup late systematically (same CU / position every step) or at random?  lateness = published - step median
        "lateness of 'published' per workgroup: std over workgroups of the per-workgroup MEAN / mean over workgroups of the per-workgroup STD": [
            float((published - np.median(published, axis=0)).mean(1).std()), float((published - np.median(published, axis=0)).std(1).mean())],
        "per-workgroup mean lateness in block order (us)": [round(float(x), 2) for x in (published - np.median(published, axis=0)).mean(1)],
        "per-workgroup mean MLP duration in block order (us)": [round(float(x), 2) for x in (st[:, 1:H - 1, 0] - built[:, :-1]).mean(1)],
        "per-workgroup mean lateness, sorted (us)": [round(float(x), 2) for x in np.sort((published - np.median(published, axis=0)).mean(1))[::10]],
        "MLP duration (built(t-1) -> MLP done(t)): mean, std over workgroups of per-workgroup mean, mean per-workgroup std over steps": [
            float((st[:, 1:H - 1, 0] - built[:, :-1]).mean()), float((st[:, 1:H - 1, 0] - built[:, :-1]).mean(1).std()), float((st[:, 1:H - 1, 0] - built[:, :-1]).std(1).mean())],
    })
print(json.dumps({"lib": os.environ.get("HIPETS_LIB", "default"), "pop": pop, "runs": res}))
