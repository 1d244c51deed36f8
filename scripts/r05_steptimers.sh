#!/bin/bash
# in-kernel stopwatch of the small multi-workgroup PROCESS steps (build with -DKOMB_STEP_TIMERS) on the alpha = 2.2 shape and C3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export KOMB_ACCEL_LIB=komb_amd/libv/timers/libkomb_accel.so
timeout -k 10 300 python3 tests/manual/scale_3x.py 10000000 27500000 2.2 > gpurun_out/steptimers_a22.txt 2>&1 || { tail -5 gpurun_out/steptimers_a22.txt; exit 1; }
grep -i "step timers" gpurun_out/steptimers_a22.txt | cut -c1-300
