#!/bin/bash
# the hand-over threshold of the k-truss local finish swept at C3 and C2 (scripts/tail_sweep.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
S=";LOCAL_LIMIT=12000000;LOCAL_LIMIT=6000000;LOCAL_LIMIT=3128767;LOCAL_LIMIT=1500000;LOCAL_LIMIT=800000;LOCAL_LIMIT=400000;LOCAL_LIMIT=200000;LOCAL_LIMIT=100000"
timeout -k 10 300 python3 scripts/tail_sweep.py 10000000 24250000 2.6 "default$S" > gpurun_out/handover_c3.txt 2>&1 || { tail -5 gpurun_out/handover_c3.txt; exit 1; }
cut -c1-200 gpurun_out/handover_c3.txt
S2=";LOCAL_LIMIT=1200000;LOCAL_LIMIT=600000;LOCAL_LIMIT=315651;LOCAL_LIMIT=150000;LOCAL_LIMIT=80000;LOCAL_LIMIT=40000;LOCAL_LIMIT=20000"
timeout -k 10 300 python3 scripts/tail_sweep.py 1000000 2450000 2.6 "default$S2" > gpurun_out/handover_c2.txt 2>&1 || { tail -5 gpurun_out/handover_c2.txt; exit 1; }
cut -c1-200 gpurun_out/handover_c2.txt
