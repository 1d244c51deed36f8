#!/bin/bash
# rocprofv3 passes of the default bench command for one config: kernel trace + stats, FETCH_SIZE, WRITE_SIZE, SQ counters
# (separate passes; nothing but the program itself after "--").  usage: gpu_profile.sh <tag> <config>
tag=$1; cfg=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/profiles
export KOMB_PROF_OUT=gpurun_out/profiles
cmd="python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-build --no-extras"
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_kt -- $cmd > gpurun_out/${tag}_kt.log 2>&1 || exit 1
timeout -k 10 250 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- $cmd > gpurun_out/${tag}_fetch.log 2>&1 || exit 1
timeout -k 10 250 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- $cmd > gpurun_out/${tag}_write.log 2>&1 || exit 1
timeout -k 10 250 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d gpurun_out/${tag}_sq -- $cmd > gpurun_out/${tag}_sq.log 2>&1 || exit 1
python3 scripts/prof_summary.py $tag gpurun_out/${tag}_kt gpurun_out/${tag}_fetch gpurun_out/${tag}_write gpurun_out/${tag}_sq > gpurun_out/${tag}_summary.txt 2>&1
# the raw CSVs are large: keep only the summaries
rm -rf gpurun_out/${tag}_kt gpurun_out/${tag}_fetch gpurun_out/${tag}_write gpurun_out/${tag}_sq
tail -n 5 gpurun_out/${tag}_summary.txt
