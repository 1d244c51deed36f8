#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 200 python3 scripts/prep_probe.py c3 2 > gpurun_out/prep_probe_c3.txt 2>&1 || { tail -5 gpurun_out/prep_probe_c3.txt; exit 1; }
grep -E "cold step 1|resident step 1|sha256" gpurun_out/prep_probe_c3.txt
for lib in default sm_16_16 sm_64_64; do
  if [ $lib = default ]; then unset KOMB_ACCEL_LIB; else export KOMB_ACCEL_LIB=komb_amd/libv/$lib/libkomb_accel.so; fi
  timeout -k 10 400 python3 tests/manual/shape_sweep.py > gpurun_out/shape_$lib.txt 2>&1 || { tail -5 gpurun_out/shape_$lib.txt; exit 1; }
  echo "== $lib"; cut -c1-250 gpurun_out/shape_$lib.txt | grep -v "^$" | tail -8
done
unset KOMB_ACCEL_LIB
