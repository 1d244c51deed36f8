# A/B of the peel driver's batch size (tuning variants built with make OUT=../libv/bN EXTRA=-DKOMB_PEEL_BATCH=N)
for cfg in c2 c3; do
for v in "" b8 b12 "" b8 b12; do
  if [ -n "$v" ]; then export KOMB_ACCEL_LIB=$PWD/komb_amd/libv/$v/libkomb_accel.so; else unset KOMB_ACCEL_LIB; fi
  python tests/manual/core_ab.py $cfg 2>/dev/null | tail -1 | sed "s/^/$v /"
  python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-build 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  $v $cfg truss', round(d['ms_per_step'],2), 'peel', round(d['phases_ms']['ms_peel'],2), 'launches', d['config']['truss_launches'])"
done
done
