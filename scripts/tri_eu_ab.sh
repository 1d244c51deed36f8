# A/B of the enumeration's occupancy (variants built with make OUT=../libv/<n> EXTRA="-DKOMB_TRI_EU=<k> -DKOMB_TRI_REC=<r>")
for v in "" rec256 eu5 "" rec256 eu5; do
  if [ -n "$v" ]; then export KOMB_ACCEL_LIB=$PWD/komb_amd/libv/$v/libkomb_accel.so; else unset KOMB_ACCEL_LIB; fi
  timeout -k 5 120 python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-build 2>/dev/null | python3 -c "
import json,sys
t=sys.stdin.read()
d=json.loads(t) if t.strip() else None
print('[$v]', 'FAILED' if d is None else (round(d['ms_per_step'],2), 'tri', round(d['phases_ms']['ms_tri_fill'],2), 'records', d['config']['tri_records']))"
done
KOMB_ACCEL_LIB=$PWD/komb_amd/libv/eu5/libkomb_accel.so timeout -k 5 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "generated or golden or threshold or cliques" 2>&1 | tail -1
