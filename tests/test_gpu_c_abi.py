"""The C-ABI boundary driven from plain C (tests/c_abi/plan_from_c.c, built with gcc by __graft_entry__.build()): no Python,
no torch in the process -- known-answer rollouts in EXACT and FAST mode and one fused CEM plan."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_plain_c_client_runs_known_answer_rollout_and_plan():
    exe = os.path.join(ROOT, "tests", "c_abi", "plan_from_c")
    assert os.path.exists(exe), "run python __graft_entry__.py first (builds the C client)"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c_abi ok" in r.stdout
