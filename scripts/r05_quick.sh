#!/bin/bash
# quick check of a change: the parity suite without the full-size cases, then the bench without the CPU baseline twice
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_komb2.py -x -q -m gpu -k "not full_size" > gpurun_out/pytest_parity.txt 2>&1; rc=$?; tail -3 gpurun_out/pytest_parity.txt; [ $rc = 0 ] || exit 1
for i in 1 2; do
timeout -k 10 300 python bench.py --no-build --no-cpu-baseline > gpurun_out/bench_nocpu_$i.json 2> gpurun_out/bench_nocpu_$i.err || { tail -5 gpurun_out/bench_nocpu_$i.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/bench_nocpu_$i.json'))
p=d['phases_ms']
print(round(d['ms_per_step'],2), round(d['ms_per_step_resident'],2), 'prep', round(p['ms_prepare'],2), 'enum', round(p['ms_tri_fill'],2), 'sort', round(p['ms_sort'],2), 'fin', round(p['ms_compact'],2), 'peel', round(p['ms_peel'],2), 'local', round(p['ms_truss_local'],2), 'gather', round(p['ms_gather'],2), '| c2 kcore', round(d['c2']['kcore']['ms'],3), 'c2 truss', round(d['c2']['ktruss']['ms_per_step'],2), round(d['c2']['ktruss']['ms_per_step_resident'],2), 'kcore', round(d['kcore']['ms'],2), 'faithful', round(d['runtruss_faithful']['ms'],2))"
done
