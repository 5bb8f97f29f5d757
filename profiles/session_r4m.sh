#!/bin/bash
# Hidden-static instances for hid 128 / 256: bitwise tests against the generic kernel, the oracle cases that now run them, and the gain
# at a cfg2-sized batch.  bash profiles/session_r4m.sh
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r4m
mkdir -p $OUT
( time timeout 600 python -m pytest tests/test_gpu_rollout.py -m gpu -q -x -p no:cacheprovider \
    -k "other_widths or kernel_class or hid256 or hid512 or hid33" --durations=5 ) > $OUT/tests.log 2>&1
echo "tests rc=$? $(tail -n 3 $OUT/tests.log | tr '\n' ' ')"
python - > $OUT/widths.json 2> $OUT/widths.err <<'PY'
import json, sys, time
sys.path[:0] = [".", "mbrl-lib_amd", "tests"]
import numpy as np, torch, hipets
from conftest import to_spec
from oracle import pets_oracle as po
dev = torch.device("cuda:0"); eng = hipets.get_engine(dev); out = {}
for hid in (128, 200, 256):
    om = po.make_synthetic_model(17, 6, ensemble_size=5, hid=hid, seed=0, termination="walker2d")  # a model without a fused instance
    eng.set_model(to_spec(om, 17, 6))
    fl = 2 * sum(int(w.shape[1]) * int(w.shape[2]) for w in om.weights)
    acts = (torch.rand(500, 30, 6, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
    s0 = (np.random.default_rng(0).standard_normal(17) * 0.1).astype(np.float32)
    res = {"kernel_class": {m: list(eng.kernel_class(500, 20, 30, m)) for m in ("device", "fast")}}
    for mode in ("device", "fast"):
        for name, kw in (("default", {}), ("generic_kernel", dict(generic_kernel=True))):
            f = lambda: eng.rollout(acts, s0, 20, mode=mode, seed=1, **kw)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3: f(); torch.cuda.synchronize()
            eng.timing_enable(True)
            for _ in range(10): f()
            torch.cuda.synchronize()
            n, ms = eng.timing_read(); eng.timing_enable(False)
            res.setdefault(mode, {})[name] = {"rollout_kernel_ms": ms / 10, "frac_of_fp32_peak": 500 * 20 * 30 * fl / (ms / 10 * 1e-3) / 157.3e12}
    out[f"hid{hid}"] = res
print(json.dumps(out, indent=1))
PY
python -c "
import json; d=json.load(open('gpurun_out/r4m/widths.json'))
for k,v in d.items(): print(k, v['kernel_class'], {m:{n:(round(x['rollout_kernel_ms'],4), round(x['frac_of_fp32_peak'],3)) for n,x in v[m].items()} for m in ('device','fast')})
"
