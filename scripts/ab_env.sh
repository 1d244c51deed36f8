#!/bin/bash
# same-box A/B of environment settings ("VAR=val,VAR2=val2" per argument; "-" = none): phases of the default bench at C3
for spec in "$@"; do
  envs=(); [ "$spec" != "-" ] && IFS=',' read -ra envs <<< "$spec"
  env "${envs[@]}" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-build 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); p=d['phases_ms']; print('$spec', round(d['ms_per_step'],2), 'fill', round(p['ms_tri_fill'],3), 'sort', round(p['ms_sort'],2), 'compact', round(p['ms_compact'],2), 'peel', round(p['ms_peel'],2), 'local', round(p['ms_truss_local'],2), 'gather', round(p['ms_gather'],2))"
done
