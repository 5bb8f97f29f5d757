"""SURVEY.md section 5 (sanitizers): the DEBUG builds of the library (__graft_entry__.build_debug, in-tree next to the shipped one;
rebuilt here only if they do not match the sources).
  libhipets_debug.so  -O1 -g, every LDS section of the rollout kernel and the indexed LDS accesses of its elementwise phases
                      bound-checked on the device (rollout.hpp HIPETS_BOUND: a violated bound prints and traps the kernel):
                      runs the driver's smoke() -- EXACT-mode rollout against the oracle + a fused CEM plan.
  libhipets_asan.so   the same, with the host side of the C ABI (hipets.hip) under AddressSanitizer: runs the plain-C client of
                      the ABI (create / set_model / EXACT + FAST rollouts / fused plan / destroy) compiled with the sanitizer too.
(The sanitizer runtime intercepts HSA allocations and fails inside the HIP runtime PyTorch bundles -- measured on the GPU box --
so the Python host runs the bounds-checked library, the C host the sanitized one.)"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_smoke_passes_on_the_bounds_checked_build():
    import __graft_entry__ as ge

    lib = ge.build_debug()
    env = dict(os.environ, HIPETS_LIB=lib)
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "[smoke]" in r.stdout and os.path.basename(lib) in r.stdout and "STALE" not in r.stdout, out[-3000:]
    assert "bound violated" not in out, out[-3000:]


def test_plain_c_client_passes_on_the_asan_build(tmp_path):
    """tests/c_abi/plan_from_c.c compiled by the same clang with -fsanitize=address and linked against libhipets_asan.so."""
    import __graft_entry__ as ge

    ge.build_debug()
    lib, rt = ge.ASAN_LIB, ge.asan_runtime()
    assert os.path.exists(lib) and os.path.exists(rt), (lib, rt)
    exe = str(tmp_path / "plan_from_c_asan")
    clang = os.path.join(os.path.dirname(ge.HIPCC), "..", "lib", "llvm", "bin", "clang")
    cmd = [clang, "-std=c99", "-O1", "-g", "-fsanitize=address", "-shared-libsan", ge.C_CLIENT_SRC, "-I" + os.path.join(ROOT, "include"),
           "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", lib, "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath," + os.path.dirname(rt), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:exitcode=23")
    r = subprocess.run([exe], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "c_abi ok" in r.stdout, out[-3000:]
    assert "AddressSanitizer" not in out and "bound violated" not in out, out[-3000:]
