#!/bin/bash
# debug: the world-2 GPU worker under several switches, three repetitions each
mkdir -p gpurun_out
port=29700
for cfg in "" "KOMB_AUX=0" "KOMB_NULL_STREAM=1" "KOMB_SCAN_SCALAR=1" "KOMB_NULL_STREAM=1 KOMB_SCAN_SCALAR=1" "KOMB_FINISH=none"; do
  ok=0
  for i in 1 2 3; do
    port=$((port+1))
    env $cfg MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=2 timeout -k 10 60 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port $port tests/_dist_worker.py gpu > gpurun_out/r02_dm.log 2>&1 && ok=$((ok+1))
  done
  echo "[$cfg] ok=$ok/3: $(grep -E 'KombError|AssertionError' gpurun_out/r02_dm.log | head -2 | tr '\n' ' ')"
done
