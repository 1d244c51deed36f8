for v in "" b8 b12; do
  for fin in local none; do
    if [ -n "$v" ]; then export KOMB_ACCEL_LIB=$PWD/komb_amd/libv/$v/libkomb_accel.so; else unset KOMB_ACCEL_LIB; fi
    KOMB_FINISH=$fin python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-build 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', '$fin', round(d['ms_per_step'],2), 'peel', round(d['phases_ms']['ms_peel'],2), 'kcore', round(d['kcore']['ms'],2), 'launches', d['config']['truss_launches'])"
  done
done
