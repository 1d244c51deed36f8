#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python3 - > gpurun_out/first_call_debug.txt 2>&1 <<'P'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import numpy as np
import komb_amd, bench
nv, ncl, alpha, seed = bench.CONFIGS["c3"][:4]
uv = komb_amd.gen_hug_edges(nv, ncl, alpha, seed)
a = komb_amd.KombAccel(); a.from_edges(nv, uv)
for i in range(3): a.truss_unprepare(); a.truss_run()
a.core_run()
deg, core = a.core_fetch()
mask = (core == core.max()).astype(np.uint8)
for i in range(3): a.truss_run(mask)
print("main ctx warmed", flush=True)
for rep in range(3):
    with komb_amd.KombAccel() as f:
        t0 = time.perf_counter(); f.from_edges(nv, uv); tb = (time.perf_counter() - t0) * 1e3
        f.set_option("POOL_DEBUG", "1")
        t0 = time.perf_counter(); f.core_run(); tc = (time.perf_counter() - t0) * 1e3
        for i in range(2):
            t0 = time.perf_counter(); f.truss_run(); t = (time.perf_counter() - t0) * 1e3
            st = f.stats()
            print(f"rep {rep}: build {tb:.1f} core {tc:.1f}; truss call {i}: {t:.1f} ms wall; prepare {st['ms_prepare']:.2f} retries {st['stream_retries']} phases tri {st['ms_tri_fill']:.2f} sort {st['ms_sort']:.2f} compact {st['ms_compact']:.2f} peel {st['ms_peel']:.2f} gather {st['ms_gather']:.2f}", flush=True)
P
tail -60 gpurun_out/first_call_debug.txt | cut -c1-250
