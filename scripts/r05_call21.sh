#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export KOMB_ACCEL_LIB=komb_amd/libv/dbg/libkomb_accel.so
KOMB_FINISH=local KOMB_LOCAL_DEBUG=1 timeout -k 10 500 python3 scripts/peel_trace.py 10000000 27500000 2.2 > gpurun_out/trace_a22.log 2>&1 || { tail -5 gpurun_out/trace_a22.log; exit 1; }
python3 scripts/peel_trace_report.py gpurun_out/peel_trace_10000000_27500000_2.2.txt > gpurun_out/peel_trace_a22_report.txt 2>&1
tail -30 gpurun_out/trace_a22.log; cat gpurun_out/peel_trace_a22_report.txt
python3 - <<'P'
import numpy as np
rows=[]; sect=[]
for ln in open('gpurun_out/peel_trace_10000000_27500000_2.2.txt'):
    if ln.startswith('#'): sect.append(len(rows)); continue
    rows.append([float(x) for x in ln.split()])
rows=np.array(rows); tr=rows[sect[0]:sect[1]]
# by level decile: steps, time, remaining
L=tr[:,1]
for lo,hi in ((0,5),(5,10),(10,20),(20,40),(40,80),(80,160),(160,400)):
    m=(L>=lo)&(L<hi)
    if m.sum(): print(f'levels [{lo},{hi}): steps {m.sum()} time {tr[m][:,8].sum()/1e3:.1f} ms, rem at start {tr[m][0,7]:.0f} live {tr[m][0,6]:.0f}; avg light {tr[m][:,3].mean():.0f} heavy {tr[m][:,4].mean():.0f}')
P
rm -f gpurun_out/peel_trace_10000000_27500000_2.2.txt
