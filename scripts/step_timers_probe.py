"""Build with EXTRA=-DKOMB_STEP_TIMERS (make OUT=../libv/tm ...), run with KOMB_ACCEL_LIB pointing at it: the peel engine prints, per
k-truss run, where block 0 spends a small multi-workgroup PROCESS step (alpha = 2.1, 1 M vertices: 2 470 such steps;
round 4: ctrl 1.1 us, queue + slice 3.6, items 5.1, flush 0.5, end-of-step barrier 5.1, ticket 0.9)."""
import os, sys
sys.path.insert(0, os.getcwd())
import komb_amd
uv = komb_amd.gen_hug_edges(1000000, 2450000, 2.1, 5)
a = komb_amd.KombAccel(); a.from_edges(1000000, uv)
for i in range(3):
    a.truss_run(); st=a.stats(); print(st['ms_peel'], st['truss_subrounds'], st['truss_launches'], flush=True)
