"""Where a (step, turn) of the TURN-BASED persistent DEVICE form goes (batches with more logical workgroups than the chip holds:
cfg4 obs 45: 435 logical workgroups of 3 row tiles in 2 turns; cfg4' obs 376: 660 of 2 row tiles in 3 turns).  Needs the
-DHIPETS_STEP_TRACE build (see handover_trace.py) selected with HIPETS_LIB.  Every launched workgroup stamps the chip-wide 100 MHz
clock at four points of every (step, turn): MLP + tail done, published, the next turn's rows arrived, the next turn's input built."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mbrl-lib_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import hipets  # noqa: E402

dev = torch.device("cuda:0")
eng = hipets.get_engine(dev)
out = {"lib": hipets.LIB_PATH}
CASES = (("cfg4p_obs376", 376, 2, 3, 1036), ("cfg4_obs45", 45, 3, 2, 1036))
if os.environ.get("TRACE_CASES"):  # "name:obs:R:turns:pop,..." -- e.g. the ragged last turn of cfg4''s fourth iCEM iteration: cfg4p_pop497:376:2:2:497
    CASES = tuple((c.split(":")[0], *[int(x) for x in c.split(":")[1:]]) for c in os.environ["TRACE_CASES"].split(","))
for name, obs, R, serve, pop in CASES:
    spec = bench.synthetic_spec(dev, obs=obs, act=17, ensemble=7, elite=[0, 1, 2, 3, 4], termination="humanoid")
    eng.set_model(spec)
    P, H = 20, 40
    acts = (torch.rand(pop, H, 17, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
    s0 = np.zeros(obs, np.float32)
    s0[0] = 1.4
    NWG, nseq, stride = 256, H * serve, H * 4  # the kernel's record stride allows up to 4 turns per step
    buf = torch.zeros(128 + NWG * stride * 4, dtype=torch.int64, device=dev)
    for i in range(10):
        eng.rollout(acts, s0, P, mode="device", seed=1, stream_id=i)
    torch.cuda.synchronize()
    buf.zero_()
    eng.rollout(acts, s0, P, mode="device", seed=1, stream_id=99, phase_cycles=buf[:128].view(8, 16))
    torch.cuda.synchronize()
    st = buf[128:].view(NWG, stride, 4).cpu().numpy().astype(np.float64)[:, :nseq] * 0.01  # us
    # workgroups that serve `serve` logical workgroups every step have a stamp in every record
    full = (st[:, :, 0] > 0).all(axis=1) & (st[:, :-1, 3] > 0).all(axis=1)
    a = st[full]
    done, pub, arr, built = (a[:, :, k] for k in range(4))
    turn = np.arange(nseq) % serve
    rec = {"launched_workgroups_with_all_turns": int(full.sum()), "records_per_workgroup": nseq}
    mlp = done[:, 1:] - built[:, :-1]  # input built (previous record) -> MLP + tail done
    rec["mlp_plus_tail_us (built -> done) by turn of the step"] = [float(mlp[:, (turn[1:] == k)].mean()) for k in range(serve)]
    rec["publish_us (done -> published)"] = float((pub - done).mean())
    wait = arr[:, :-1] - pub[:, :-1]
    rec["collect_us (published -> next turn's rows arrived) by the turn that follows"] = [float(wait[:, (turn[1:] == k)].mean()) for k in range(serve)]
    bld = built[:, :-1] - arr[:, :-1]
    rec["input_build_us (arrived -> built)"] = float(bld.mean())
    rec["record_to_record_us (done -> done) by turn"] = [float(np.diff(done, axis=1)[:, (turn[1:] == k)].mean()) for k in range(serve)]
    rec["step_us"] = float(np.diff(done[:, ::serve], axis=1).mean())
    rec["rollout_ms_from_stamps"] = float((done.max() - done.min()) / 1e3)
    out[name] = rec
print(json.dumps(out, indent=1))
