#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python3 scripts/anomaly_probe.py > gpurun_out/anomaly_probe.txt 2>&1; cat gpurun_out/anomaly_probe.txt | tail -30
