#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 400 python3 scripts/anomaly_probe2.py > gpurun_out/anomaly_probe2.txt 2>&1; tail -40 gpurun_out/anomaly_probe2.txt
