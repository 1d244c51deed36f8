#!/bin/bash
# runs beyond the bench configuration on the final build: 3 x C3, alpha = 2.2 at C3 size, the maximum-size graph (> 2^31 CSR slots)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python3 tests/manual/scale_3x.py > gpurun_out/r05_scale_3x.txt 2>&1 || { tail -5 gpurun_out/r05_scale_3x.txt; exit 1; }
tail -3 gpurun_out/r05_scale_3x.txt | cut -c1-300
timeout -k 10 300 python3 tests/manual/scale_3x.py 10000000 27500000 2.2 > gpurun_out/r05_alpha22_c3size.txt 2>&1 || { tail -5 gpurun_out/r05_alpha22_c3size.txt; exit 1; }
tail -3 gpurun_out/r05_alpha22_c3size.txt | cut -c1-300
timeout -k 10 500 python3 tests/manual/big_raw.py > gpurun_out/r05_big_raw.txt 2>&1 || { tail -8 gpurun_out/r05_big_raw.txt; exit 1; }
tail -6 gpurun_out/r05_big_raw.txt | cut -c1-400
