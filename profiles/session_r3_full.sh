#!/bin/bash
# round 3 evidence session: profiles/session_full.sh r3 (smoke, the whole GPU test suite, default bench line, gloo2 bench, rocprofv3
# collection) + the round's extra probes (k-loop probe, phase profile of the shape-specialised kernel, hand-over trace).
# The probes need two variant builds of the library next to the shipped one (built HERE before the session, removed afterwards so
# that only libhipets.so travels with the tree):
#   python -c "import __graft_entry__ as g, os; [g.build_library(out=os.path.join(g.PKG,'hipets','libhipets_%s.so'%t), extra_flags=(f,), \
#              objdir=os.path.join(g.PKG,'build_'+t), force=True) for t,f in (('prof','-DHIPETS_LEAN_PROF=1'),('trace','-DHIPETS_STEP_TRACE'))]"
set -u
export TMPDIR=/tmp
bash profiles/session_full.sh r3
OUT=gpurun_out/full_r3
profiles/microbench/kloop_probe > $OUT/kloop_probe.jsonl 2>&1
HIPETS_LIB=$PWD/mbrl-lib_amd/hipets/libhipets_prof.so python profiles/kernel_variants.py 2>/dev/null | grep '^{' | tail -1 > $OUT/kernel_variants_prof.json
python profiles/kernel_variants.py 2>/dev/null | grep '^{' | tail -1 > $OUT/kernel_variants.json
HIPETS_LIB=$PWD/mbrl-lib_amd/hipets/libhipets_trace.so python profiles/handover_trace.py 2>/dev/null | grep '^{' | tail -1 > $OUT/handover_trace.json
HIPETS_LIB=$PWD/mbrl-lib_amd/hipets/libhipets_prof.so python profiles/small_batch_probe.py 2>/dev/null | grep '^{' | tail -1 > $OUT/small_batches_prof.json
echo all done
