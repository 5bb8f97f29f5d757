"""Build a VARIANT of libhipets.so for A/B measurements: the named translation units recompiled with extra -D flags, every other
object taken from the shipped build (mbrl-lib_amd/build/), linked into profiles/variants/<name>.so (git-ignored: *.so; it travels
to the GPU box with the snapshot).  HIPETS_LIB=<that path> selects it (mbrl-lib_amd/hipets/_lib.py).  The ISA hazard scan of
__graft_entry__ runs on every recompiled unit: a variant with findings is not linked.

    python profiles/build_variant.py noks rollout_r1.hip -DHIPETS_KSPLIT=0
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    name, rest = sys.argv[1], sys.argv[2:]
    units = [a for a in rest if a.endswith(".hip")]
    flags = [a for a in rest if not a.endswith(".hip")]
    ge.build_library()  # the shipped objects must be current
    objdir = os.path.join(ROOT, "profiles", "variants", name + "_obj")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for u in ge.UNITS:
        if u not in units:
            objs.append(os.path.join(ge.OBJDIR, u.replace(".hip", ".o")))
            continue
        o = os.path.join(objdir, u.replace(".hip", ".o"))
        cmd = [ge.HIPCC] + ge.FLAGS + flags + ["-save-temps=obj", "-c", os.path.join(ge.CSRC, u), "-o", o]
        print("[variant]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        for f in os.listdir(objdir):
            path = os.path.join(objdir, f)
            if f.endswith("gfx950.s"):
                n, found = ge.scan_isa_hazards(path)
                res = ge.parse_kernel_resources(path)
                scratch = {k: v for k, v in res.items() if v.get("ScratchSize [bytes/lane]", 0)}
                print(f"[variant] {u}: {n} asm MFMAs, {len(found)} hazard finding(s), {len(scratch)} kernel(s) with scratch")
                if found and not os.environ.get("HIPETS_ALLOW_ISA_HAZARDS"):
                    raise SystemExit("\n".join(found[:10]))
            if not (f.endswith(".o") and f[:-2] + ".hip" in units):  # -save-temps files; the recompiled units' objects stay
                os.remove(path)
        objs.append(o)
    out = os.path.join(ROOT, "profiles", "variants", name + ".so")
    subprocess.run([ge.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out], check=True)
    print(out)


if __name__ == "__main__":
    main()
