#!/bin/bash
# C3 step under different item limits of the k-truss local finish (KOMB_LOCAL_ITEMS); same box, one call
for it in default 16000000 48000000 64000000 96000000 128000000 default; do
  if [ "$it" = default ]; then unset KOMB_LOCAL_ITEMS; else export KOMB_LOCAL_ITEMS=$it; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-build 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); p=d['phases_ms']; c=d['config']; print('items=$it', round(d['ms_per_step'],2), 'peel', round(p['ms_peel'],2), 'local', round(p['ms_truss_local'],2), c['truss_local'], 'kcore', round(d['kcore']['ms'],2))"
done
