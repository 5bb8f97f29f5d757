#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r4j; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-900} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run tests python -m pytest -m gpu -q --maxfail=10 -p no:cacheprovider tests/test_gpu_rollout.py tests/test_gpu_device_mode.py tests/test_gpu_closed_forms.py tests/test_gpu_bf16x3.py
python profiles/cfg4p_probe.py > $OUT/cfg4p.json 2> $OUT/err1.log; cat $OUT/cfg4p.json
python profiles/other_configs.py --reps 6 > $OUT/other_configs.json 2> $OUT/err2.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4j/other_configs.json'))
for k,v in d.items():
    print(k, {m:{a:round(b,4) for a,b in x.items() if isinstance(b,float)} for m,x in v.items() if isinstance(x,dict)})
PY
echo done
