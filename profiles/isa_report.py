"""Static instruction statistics of rollout-kernel instances from the final ISA: how many register-spill moves (v_readlane / v_writelane
for SGPR spills, v_accvgpr_read / _write for VGPRs parked in accumulation registers), MFMAs, LDS and memory instructions a kernel
contains, next to the compiler's resource notes.  The kernels' step loops are fully unrolled (one copy of every layer's k loop per
instance), so an instruction outside the few small loops executes at most once per step: the static counts bound the per-step counts.
Compiles one translation unit with -save-temps into profiles/variants/isa_obj/ (git-ignored).

    python profiles/isa_report.py rollout_r3.hip 'KSpec<1, 13, 3, 2, 0, 4, 0, 0, 0, 1>' 'KSpec<1, 13, 3, 2, 0, 4, 0, 1, 0, 1>' > profiles/r5_isa_report.json
"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

CLASSES = [("mfma", r"^v_mfma"), ("sgpr_spill_moves (v_readlane / v_writelane)", r"^v_(read|write)lane_b32"),
           ("accvgpr_moves (v_accvgpr_read / _write)", r"^v_accvgpr_(read|write)"), ("scratch", r"^scratch_"), ("lds", r"^ds_"),
           ("vmem_load", r"^(buffer|global)_load"), ("vmem_store", r"^(buffer|global)_store"), ("waitcnt", r"^s_waitcnt"),
           ("barrier", r"^s_barrier"), ("branch", r"^s_c?branch"), ("f64_valu", r"^v_\w+_f64"), ("salu", r"^s_")]


def main():
    unit, wanted = sys.argv[1], sys.argv[2:]
    objdir = os.path.join(ROOT, "profiles", "variants", "isa_obj")
    os.makedirs(objdir, exist_ok=True)
    o = os.path.join(objdir, unit.replace(".hip", ".o"))
    subprocess.run([ge.HIPCC] + ge.FLAGS + ["-save-temps=obj", "-c", os.path.join(ge.CSRC, unit), "-o", o], check=True, stderr=subprocess.DEVNULL)
    asm = [os.path.join(objdir, f) for f in os.listdir(objdir) if f.endswith("gfx950.s")][0]
    res = ge.parse_kernel_resources(asm)
    out = {"unit": unit, "flags": " ".join(ge.FLAGS), "kernels": {}}
    cur, counts = None, None
    demangled = {}
    names = [k for k in res]
    dm = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
    demangled = dict(zip(names, dm))
    for line in open(asm, errors="replace"):
        m = re.match(r"^(\S+):\s+; @", line)
        if m:
            cur = m.group(1) if m.group(1) in res else None
            counts = collections.Counter()
            if cur:
                out["kernels"][cur] = counts
            continue
        if cur is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
            continue
        t = line.strip()
        if not t or t[0] in ".;" or t.endswith(":"):
            continue
        op = t.split()[0]
        counts["instructions"] += 1
        for name, rx in CLASSES:
            if re.match(rx, op):
                counts[name] += 1
                break
    final = {}
    for k, c in out["kernels"].items():
        d = demangled.get(k, k)
        if wanted and not any(w in d for w in wanted):
            continue
        final[d] = {"static instruction counts": dict(c), "compiler notes": res[k]}
    out["kernels"] = final
    print(json.dumps(out, indent=1))
    for f in os.listdir(objdir):
        os.remove(os.path.join(objdir, f))


if __name__ == "__main__":
    main()
