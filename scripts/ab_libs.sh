#!/bin/bash
# same-box A/B of library builds: ab_libs.sh <rounds> <name=path> ...  (bench.py C3, 10 steps; prints step and phase times)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    KOMB_ACCEL_LIB=$lib timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-build 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_ms']
print('$name step %.2f orient %.2f tri %.2f sort %.2f compact %.2f peel %.2f local %.2f gather %.2f'%(d['ms_per_step'],p['ms_orient'],p['ms_tri_fill'],p['ms_sort'],p['ms_compact'],p['ms_peel'],p['ms_truss_local'],p['ms_gather']))"
  done
done
