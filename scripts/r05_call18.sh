#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for lib in default pab1 pab2 pab4 pab7; do
  if [ $lib = default ]; then unset KOMB_ACCEL_LIB; else export KOMB_ACCEL_LIB=komb_amd/libv/$lib/libkomb_accel.so; fi
  timeout -k 10 200 python3 scripts/prep_probe.py c3 2 > gpurun_out/prep_probe_$lib.txt 2>&1 || { tail -5 gpurun_out/prep_probe_$lib.txt; }
  echo "== $lib: $(grep 'cold step 1' gpurun_out/prep_probe_$lib.txt | cut -c1-120)"
done
unset KOMB_ACCEL_LIB
timeout -k 10 400 python3 tests/manual/scale_3x.py > gpurun_out/r05_scale_3x.txt 2>&1; tail -12 gpurun_out/r05_scale_3x.txt | cut -c1-400
timeout -k 10 500 python3 tests/manual/scale_3x.py 10000000 27500000 2.2 > gpurun_out/r05_alpha22_c3size.txt 2>&1; tail -12 gpurun_out/r05_alpha22_c3size.txt | cut -c1-400
