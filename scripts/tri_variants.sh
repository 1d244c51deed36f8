#!/bin/bash
# bench.py phases for several builds of the library (komb_amd/libv/<name>/libkomb_accel.so, see Makefile OUT=/EXTRA=)
# usage (GPU box): tri_variants.sh <config> name...
cfg=$1; shift
for n in "$@"; do
    lib=komb_amd/lib/libkomb_accel.so
    [ "$n" != base ] && lib=komb_amd/libv/$n/libkomb_accel.so
    echo "== $n"
    KOMB_ACCEL_LIB=$PWD/$lib python bench.py --no-build --config $cfg --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d.get('phases_ms'), d['roofline'].get('kernel'), d['roofline'].get('frac'))
"
done
