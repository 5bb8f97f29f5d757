set -u
export TMPDIR=/tmp
OUT=gpurun_out/r3h; mkdir -p $OUT
for v in main noil nounroll; do
  if [ $v = main ]; then unset HIPETS_LIB; else export HIPETS_LIB=$PWD/mbrl-lib_amd/hipets/libhipets_$v.so; fi
  python -m pytest -m gpu -q -p no:cacheprovider tests/test_gpu_rollout.py -k "pop2000 and (fast_mode_replayed or shape_specialised)" > $OUT/t_$v.log 2>&1
  echo "$v: $(tail -1 $OUT/t_$v.log)"
done
