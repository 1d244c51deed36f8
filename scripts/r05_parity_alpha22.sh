#!/bin/bash
# heavy-tailed graph (alpha = 2.2, |V| = 5 M): every coreness, support and trussness value against the oracle
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1150 python3 tests/manual/c3_parity_oneoff.py 5000000 13750000 2.2 16 > gpurun_out/r05_parity_alpha22_5M_vertices.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/r05_parity_alpha22_5M_vertices.log
