"""The C-ABI library loads and exports every symbol include/hipets.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "hipets.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hipets_[a-z_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    fns = header_functions()
    for must in ("hipets_create", "hipets_destroy", "hipets_set_model", "hipets_rollout", "hipets_cem_sample",
                 "hipets_cem_refit", "hipets_plan_cem", "hipets_plan_icem", "hipets_plan_mppi", "hipets_last_error",
                 "hipets_abi_version"):
        assert must in fns


def test_library_exports_every_declared_symbol():
    from hipets import _lib

    lib = _lib.load()
    for name in header_functions():
        assert hasattr(lib, name), f"libhipets.so does not export {name}"
    assert sorted(_lib.SYMBOLS) == header_functions(), "ctypes binding and header disagree"
    assert lib.hipets_abi_version() == _lib.ABI_VERSION


def test_struct_layouts_match_header_field_order():
    """Field names of the ctypes structs follow the header's declaration order."""
    from hipets import _lib

    src = open(HEADER).read()

    def fields(struct_name):
        body = re.search(r"typedef struct \{([^{}]*)\} " + struct_name + ";", src, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        return [re.findall(r"([a-z_0-9]+)\s*$", decl.strip())[0] for decl in body.split(";") if decl.strip()]

    assert fields("hipets_model_desc") == [f[0] for f in _lib.ModelDesc._fields_]
    assert fields("hipets_rollout_opts") == [f[0] for f in _lib.RolloutOpts._fields_]
    assert fields("hipets_cem_params") == [f[0] for f in _lib.CemParams._fields_]
    assert fields("hipets_icem_params") == [f[0] for f in _lib.IcemParams._fields_]
    assert fields("hipets_planet_desc") == [f[0] for f in _lib.PlanetDesc._fields_]
    assert fields("hipets_planet_opts") == [f[0] for f in _lib.PlanetOpts._fields_]
    assert fields("hipets_plan_trace") == [f[0] for f in _lib.PlanTrace._fields_]


def test_integration_md_stub_matches_header_and_binding():
    """The reference-side ctypes stub printed in INTEGRATION.md is executed as written: every Structure it defines must have
    the header's field order and the binding's field types (a stale stub shifts every later field by 4 bytes)."""
    from hipets import _lib

    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    stub = next(b for b in blocks if "class RolloutOpts" in b)
    classes = re.findall(r"(class \w+\(C\.Structure\):.*?\n(?=\S|\Z))", stub, flags=re.S)
    assert classes, "no ctypes Structure found in the INTEGRATION.md stub"
    ns = {"C": ctypes}
    for c in classes:
        exec(c, ns)
    by_header = {"RolloutOpts": _lib.RolloutOpts, "ModelDesc": _lib.ModelDesc, "CemParams": _lib.CemParams,
                 "IcemParams": _lib.IcemParams, "PlanTrace": _lib.PlanTrace}
    checked = 0
    for name, ref in by_header.items():
        if name in ns:
            assert [(f[0], f[1]) for f in ns[name]._fields_] == [(f[0], f[1]) for f in ref._fields_], name
            assert ctypes.sizeof(ns[name]) == ctypes.sizeof(ref)
            checked += 1
    assert checked >= 1


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a GPU")
def test_no_silent_cpu_fallback():
    """Without a GPU the product refuses to run instead of computing on the CPU."""
    import hipets
    from hipets import _lib

    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.hipets_create(0, ctypes.byref(h)) != 0
    assert b"no CPU fallback" in lib.hipets_last_error()
    with pytest.raises(hipets.HipetsError):
        hipets.get_engine("cpu")
    with pytest.raises(hipets.HipetsError):
        hipets.CEMOptimizer(2, 0.1, 10, [[-1.0]], [[1.0]], 0.1, "cpu")


def test_product_never_imports_oracle():
    """The parity claim is void if the product path routes through the oracle: grep the package."""
    pkg = os.path.join(ROOT, "mbrl-lib_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "pets_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_header_is_plain_c_and_the_c_client_links():
    """include/hipets.h compiles as C99 with -Wall -Wextra -Werror and a C program links against libhipets.so
    (no C++ or torch types in any signature)."""
    import sys

    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge

    ge.build_c_client(force=True)
    assert os.path.exists(ge.C_CLIENT)


def test_no_kernel_of_the_shipped_library_uses_scratch_memory():
    """The compiler's own resource report of every kernel (recorded by __graft_entry__.build_library next to the library):
    zero bytes of scratch.  Register-resident fragment arrays are what the rollout kernel's k loop lives on; one dynamically
    indexed array (a loop the compiler did not unroll) moves them to scratch memory without any diagnostic -- same results,
    28 ms instead of 1 ms per rollout (round 3, caught by timing only) -- and an instantiation too many pushes a kernel that
    sits at the 256-VGPR limit into spilling.  No exception is left: until round 6 the coloured-noise sampler kept its H/2+1 spectrum
    coefficients in a run-time indexed per-thread array (scratch: 72 us per cfg4 iteration) -- its frequency loops are unrolled over
    registers now; until round 5 the two R = 2 instances of the bf16x3 arithmetic mode spilt to scratch: deleted."""
    import json
    import sys

    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge

    ge.build_library()
    info = json.load(open(ge.BUILDINFO))
    kernels = info.get("kernels", {})
    rollout = [k for k in kernels if "rollout_kernel" in k]
    assert len(rollout) >= 20, "the resource report of the rollout-kernel instances is missing from the buildinfo file"

    assert any("icem_sample_kernel" in k for k in kernels)
    for name, r in kernels.items():
        assert r.get("ScratchSize [bytes/lane]", 0) == 0, (name, r)


def test_isa_scanner_sees_both_hazards_of_an_asm_mfma(tmp_path):
    """scan_isa_hazards on hand-written snippets: a VALU write of SrcC straight in front of an asm MFMA (the pattern the compiler
    produced for the cfg4 instances: the zero of an odd-k-step accumulator sunk to its first use), a VALU read of the result
    right behind one, and the two legal forms (wait states in between / result consumed after a drain)."""
    import sys

    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge

    def scan(body):
        p = tmp_path / "k.s"
        p.write_text("kern:\n" + body)
        return ge.scan_isa_hazards(str(p))

    mfma = "\t;;#ASMSTART\n\tv_mfma_f32_16x16x4_f32 v[30:33], v37, v53, v[30:33]\n\t;;#ASMEND\n"
    n, found = scan("\tv_mov_b64_e32 v[32:33], s[70:71]\n" + mfma)
    assert n == 1 and len(found) == 1 and "writes a source" in found[0]
    n, found = scan("\tv_mov_b64_e32 v[32:33], s[70:71]\n\ts_nop 1\n" + mfma)
    assert n == 1 and found == []
    n, found = scan(mfma + "\tv_add_f32_e32 v1, v30, v2\n")
    assert len(found) == 1 and "in flight" in found[0]
    n, found = scan(mfma + "\ts_nop 15\n\tv_add_f32_e32 v1, v30, v2\n")
    assert found == []
    n, found = scan(mfma + mfma.replace("v[30:33]", "v[40:43]") + "\tv_add_f32_e32 v1, v30, v2\n")  # 32 + 4 cycles < 40
    assert len(found) == 1
    # control flow: a loop whose LAST instruction is an asm MFMA and whose FIRST reads that MFMA's result -- invisible top to bottom
    # (the read comes first in the text), found along the back-edge; the same loop with the result drained before the branch is clean
    loop = "\tv_mov_b32_e32 v9, 0\n.LBB0_1:\n\tv_add_f32_e32 v1, v30, v2\n\ts_nop 15\n" + mfma
    n, found = scan(loop + "\ts_cbranch_scc1 .LBB0_1\n\ts_endpgm\n")
    assert n == 1 and len(found) == 1 and "in flight" in found[0] and "reached by a branch to .LBB0_1" in found[0]
    n, found = scan(loop + "\ts_nop 15\n\ts_cbranch_scc1 .LBB0_1\n\ts_endpgm\n")
    assert found == []
    # ... and a VALU write of an operand in the last slot before a forward branch into an MFMA
    n, found = scan("\tv_mov_b64_e32 v[32:33], s[70:71]\n\ts_cbranch_scc0 .LBB0_2\n\ts_nop 7\n\ts_nop 7\n.LBB0_2:\n" + mfma)
    assert len(found) == 1 and "writes a source" in found[0]
    # an unconditional jump ends a fall-through path: the block behind it in the text is reached by its own branches only -- no
    # finding for "MFMA; s_branch away; <label>: read of that register" unless a branch really carries the result there
    away = mfma + "\ts_branch .LBB0_9\n.LBB0_3:\n\tv_add_f32_e32 v1, v30, v2\n.LBB0_9:\n\ts_nop 15\n\ts_endpgm\n"
    n, found = scan(away)
    assert n == 1 and found == []
    n, found = scan("\ts_cbranch_scc0 .LBB0_4\n" + away.replace(".LBB0_9\n.LBB0_3:", ".LBB0_3\n.LBB0_3:"))  # the jump now leads INTO the read
    assert len(found) == 1 and "reached by a branch to .LBB0_3" in found[0]
    # (c) an asm VMEM store of more than 8 bytes reads its data registers late: a write to one of them needs two wait states behind it
    store = "\t;;#ASMSTART\n\tglobal_store_dwordx4 v[8:9], v[4:7], off sc1\n\t;;#ASMEND\n"
    n, found = scan(store + "\ts_or_b64 exec, exec, s[2:3]\n\tv_lshl_add_u32 v4, v168, 2, s50\n")  # the sequence round 5 found in an R = 2 instance
    assert len(found) == 1 and "overwrites the data" in found[0]
    n, found = scan(store.replace("sc1\n", "sc1\n\ts_nop 1\n") + "\tv_lshl_add_u32 v4, v168, 2, s50\n")
    assert found == []
    n, found = scan(store + "\tv_mov_b32_e32 v10, v4\n\tv_add_u32_e32 v11, v5, v6\n\tv_mov_b32_e32 v4, 0\n")  # reads are free; the write is 2 states away
    assert found == []
    n, found = scan("\tglobal_store_dwordx4 v[8:9], v[4:7], off\n\tv_mov_b32_e32 v4, 0\n")  # the compiler's own store: its hazard recogniser's business
    assert found == []
    # a compiler-issued MFMA (builtin, no asm markers) is the hazard recogniser's business, not the scan's
    n, found = scan("\tv_mov_b64_e32 v[32:33], s[70:71]\n\tv_mfma_f32_16x16x4_f32 v[30:33], v37, v53, v[30:33]\n")
    assert n == 0 and found == []


def test_no_asm_mfma_of_the_shipped_library_sits_in_a_hazard():
    """Every asm MFMA of the shipped kernels (52 k of them), checked in the final ISA at build time (scan_isa_hazards; the
    findings are recorded in the buildinfo file): no instruction touches an MFMA result that is still in flight, no VALU
    write lands on an MFMA operand without its wait states.  The compiler cannot see into the asm statements; round 3 found
    accumulator initialisations it had sunk to just in front of their first MFMA (wrong sums in the cfg4 instances)."""
    import json
    import sys

    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge

    ge.build_library()
    isa = json.load(open(ge.BUILDINFO)).get("isa")
    assert isa, "the ISA scan is missing from the buildinfo file (rebuild: python __graft_entry__.py --force)"
    assert sorted(isa["units_scanned"]) == sorted(ge.UNITS)
    assert isa["asm_mfmas"] > 20000
    assert isa["hazards"] == [], isa["hazards"][:5]


def test_kernel_resources_are_read_from_the_assembly_annotations(tmp_path):
    """parse_kernel_resources on a hand-written snippet: registers / scratch / occupancy from the "; Kernel info:" block of a kernel,
    spill counts from its metadata note; functions without a note (device functions) are not reported."""
    import sys

    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge

    p = tmp_path / "k.s"
    p.write_text("helper:  ; @helper\n; NumVgprs: 12\n; ScratchSize: 0\n"
                 "kern:    ; @kern\n\ts_endpgm\n; Kernel info:\n; NumVgprs: 254\n; NumAgprs: 68\n; ScratchSize: 16\n; Occupancy: 1\n"
                 "amdhsa.kernels:\n  - .agpr_count: 68\n    .name:           kern\n    .sgpr_spill_count: 82\n    .vgpr_spill_count: 3\n")
    res = ge.parse_kernel_resources(str(p))
    assert res == {"kern": {"VGPRs": 254, "AGPRs": 68, "ScratchSize [bytes/lane]": 16, "Occupancy [waves/SIMD]": 1, "SGPRs Spill": 82, "VGPRs Spill": 3}}
