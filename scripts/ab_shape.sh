#!/bin/bash
# same-box A/B of library builds on another shape of the generator: ab_shape.sh <nv> <n_cliques> <alpha> <name=path> ...
nv=$1; nc=$2; al=$3; shift 3
for r in 1 2; do for spec in "$@"; do name=${spec%%=*}; lib=${spec#*=}
KOMB_ACCEL_LIB=$lib timeout -k 10 300 python - <<P
import komb_amd
uv = komb_amd.gen_hug_edges($nv, $nc, $al, 42)
a = komb_amd.KombAccel(); a.from_edges($nv, uv)
for i in range(3): a.truss_run(); a.core_run()
s = a.stats()
print("$name truss peel %.2f ms (%d sub-rounds, local %.2f) core %.2f ms" % (s["ms_peel"], s["truss_subrounds"], s["ms_truss_local"], s["ms_core"]))
P
done; done
