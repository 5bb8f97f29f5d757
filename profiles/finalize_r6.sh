#!/bin/bash
# build box, after profiles/session_r6_final.sh: summaries into profiles/, new memo entries (if any) into tests/golden, then the CPU suite --
# AFTER the files it reads were rewritten (round-5 verdict: the closing session once rewrote a JSON a CPU test asserted on, unseen).
set -eu
cd "$(dirname "$0")/.."
python profiles/finalize_r6.py
python profiles/summarize.py r6
if ls gpurun_out/oracle_cache/*.npz >/dev/null 2>&1; then cp gpurun_out/oracle_cache/*.npz tests/golden/oracle_cache/; fi
HIPETS_BUILD_DEBUG=0 python -m pytest tests -q -m "not gpu"
