#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for i in 1 2; do
timeout -k 10 300 python bench.py --no-build --no-cpu-baseline > gpurun_out/bench_nocpu_$i.json 2> gpurun_out/bench_nocpu_$i.err || { tail -5 gpurun_out/bench_nocpu_$i.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/bench_nocpu_$i.json'))
print(d['ms_per_step'], d['ms_per_step_resident'], json.dumps(d['first_call'])[100:], d['runtruss_faithful'], d['c2']['kcore']['ms'], d['c2']['ktruss']['ms_per_step'], d['kcore']['ms'], d['corea'])"
done
