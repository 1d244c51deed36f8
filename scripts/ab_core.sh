#!/bin/bash
# same-box A/B of library variants: k-core time at C3 and C2 (results of experiment builds may be wrong on purpose)
for n in "$@"; do
  lib=komb_amd/libv/$n/libkomb_accel.so; [ "$n" = base ] && lib=komb_amd/lib/libkomb_accel.so
  KOMB_ACCEL_LIB=$lib python - <<PY
import komb_amd, time
for nv, ncl in ((10_000_000, 24_250_000), (1_000_000, 2_450_000)):
    uv = komb_amd.gen_hug_edges(nv, ncl, 2.6, 42)
    a = komb_amd.KombAccel(); a.from_edges(nv, uv); del uv
    a.core_run(); a.core_run(); s = a.stats()
    print("$n", nv, "k-core %.2f ms (local %.2f, %d units)" % (s["ms_core"], s["ms_core_local"], s["core_local_units"]), "launches", s["core_launches"], "kmax", s["max_coreness"], flush=True)
    a.close()
PY
done
