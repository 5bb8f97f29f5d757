"""SURVEY.md section 5 (sanitizers): the DEBUG build of the library -- -O1 -g, every LDS section of the rollout kernel and the indexed
LDS accesses of its elementwise phases bound-checked on the device (rollout.hpp HIPETS_BOUND: a violated bound aborts the kernel),
the host side of the C ABI under AddressSanitizer -- runs the driver's smoke() (EXACT-mode rollout against the oracle + a fused CEM
plan) and the plain-C client of the ABI, clean.  The library is built by __graft_entry__.build_debug() (in-tree, next to the shipped
one; rebuilt here only if it does not match the sources)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _env():
    import __graft_entry__ as ge

    lib = ge.build_debug()
    rt = ge.asan_runtime()
    assert os.path.exists(rt), f"AddressSanitizer runtime not found ({rt})"
    env = dict(os.environ)
    env.update(HIPETS_LIB=lib, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=23")
    return env, lib


def test_smoke_passes_on_the_bounds_checked_asan_build():
    env, lib = _env()
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "[smoke]" in r.stdout and os.path.basename(lib) in r.stdout and "STALE" not in r.stdout, out[-3000:]
    assert "AddressSanitizer" not in out and "Assertion" not in out, out[-3000:]


def test_plain_c_client_passes_on_the_bounds_checked_asan_build(tmp_path):
    """tests/c_abi/plan_from_c.c (create / set_model / rollout / plan / destroy through the C ABI, no Python) compiled by the same
    clang with -fsanitize=address and linked against the debug library."""
    import __graft_entry__ as ge

    env, lib = _env()
    exe = str(tmp_path / "plan_from_c_asan")
    clang = os.path.join(os.path.dirname(ge.HIPCC), "..", "lib", "llvm", "bin", "clang")
    cmd = [clang, "-std=c99", "-O1", "-g", "-fsanitize=address", "-shared-libsan", ge.C_CLIENT_SRC, "-I" + os.path.join(ROOT, "include"),
           "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", lib, "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath," + os.path.dirname(ge.asan_runtime()), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    env.pop("LD_PRELOAD")  # the executable links the runtime itself
    r = subprocess.run([exe], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "AddressSanitizer" not in out, out[-3000:]
