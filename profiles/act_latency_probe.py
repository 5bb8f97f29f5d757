"""TrajectoryOptimizerAgent.act on cfg2, call by call, in a fresh process: mean / median / max over 150 synchronous calls with the
garbage collector on and off, and every collector pass longer than 1 ms (is a slow call the library or a host pause?).
    python profiles/act_latency_probe.py      (MI355X box, from the repo root)"""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mbrl-lib_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, hipets
dev = torch.device("cuda:0")
eng = hipets.get_engine(dev)
spec = bench.synthetic_spec(dev)
s0 = (np.random.default_rng(0).standard_normal(bench.OBS) * 0.1).astype(np.float32)
cfg = dict(_target_="hipets.CEMOptimizer", num_iterations=bench.ITERS, elite_ratio=bench.ELITE_RATIO, population_size=bench.POP, alpha=bench.ALPHA,
           device=dev, lower_bound="???", upper_bound="???", return_mean_elites=True, seed=0)
ag = hipets.TrajectoryOptimizerAgent(cfg, [-1.0] * bench.ACT, [1.0] * bench.ACT, planning_horizon=bench.HORIZON)
ag.set_trajectory_eval_fn(hipets.make_eval_fn(spec, bench.PARTICLES, engine=eng, seed=0, mode="device"))
def run(tag, n=150):
    for _ in range(3): ag.act(s0)
    ts = []
    for _ in range(n):
        a = time.perf_counter(); ag.act(s0); ts.append(1e3 * (time.perf_counter() - a))
    ts = np.array(ts)
    print(tag, "mean %.3f median %.3f max %.2f" % (ts.mean(), np.median(ts), ts.max()), "outliers(idx,ms):", [(int(i), round(float(ts[i]), 1)) for i in np.nonzero(ts > 7)[0]])
gcs = []
def cb(phase, info):
    if phase == "start": cb.t = time.perf_counter()
    else: gcs.append((info["generation"], round(1e3 * (time.perf_counter() - cb.t), 2)))
gc.callbacks.append(cb)
run("gc on ")
print("gc passes (gen, ms) over 1 ms:", [g for g in gcs if g[1] > 1.0])
gc.collect(); gc.disable()
run("gc off")
gc.enable()
