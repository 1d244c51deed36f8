#!/bin/bash
# same-box A/B of library variants (komb_amd/libv/<name>): enumeration time of the default bench at C3
for n in "$@"; do
  lib=komb_amd/libv/$n/libkomb_accel.so; [ "$n" = base ] && lib=komb_amd/lib/libkomb_accel.so
  KOMB_ACCEL_LIB=$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-build 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); p=d['phases_ms']; print('$n', round(d['ms_per_step'],2), 'fill', round(p['ms_tri_fill'],3), 'sort', round(p['ms_sort'],2), 'compact', round(p['ms_compact'],2), 'peel', round(p['ms_peel'],2), 'local', round(p['ms_truss_local'],2), 'gather', round(p['ms_gather'],2), 'kcore', round(d['kcore']['ms'],2), 'kcore_local', round(d['kcore']['local']['ms'],2))"
done
