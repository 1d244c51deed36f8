#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cmd="python3 bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-build --no-extras"
for v in default default; do
  if [ $v = default ]; then unset KOMB_ACCEL_LIB; else export KOMB_ACCEL_LIB=komb_amd/libv/$v/libkomb_accel.so; fi
  timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_kt -- $cmd > gpurun_out/tl_kt.log 2>&1 || { tail -5 gpurun_out/tl_kt.log; exit 1; }
  python3 scripts/timeline.py gpurun_out/tl_kt 3 > gpurun_out/timeline_c3_$v.txt 2>&1
  rm -rf gpurun_out/tl_kt
  echo "== $v"; grep -E "kernels, wall|k_prep_|k_truss_results|k_wedges|k_bin_finish " gpurun_out/timeline_c3_$v.txt | grep -v "+"
done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "preparation_long_rows or cliques_and_hubs or induced or internal or degree_order" > gpurun_out/pytest_prep.txt 2>&1; rc=$?; tail -3 gpurun_out/pytest_prep.txt; [ $rc = 0 ] || exit 1
timeout -k 10 200 python3 scripts/prep_probe.py c3 3 > gpurun_out/prep_probe_c3.txt 2>&1 || { tail -5 gpurun_out/prep_probe_c3.txt; exit 1; }
cat gpurun_out/prep_probe_c3.txt | cut -c1-220
