#!/bin/bash
# round 5, call q: does RCCL allow two ranks on one device here?
mkdir -p gpurun_out/r05q
MASTER_ADDR=127.0.0.1 timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 tools/rccl_two_ranks_one_gpu_probe.py > gpurun_out/r05q/probe.log 2>&1
echo "rc=$?" >> gpurun_out/r05q/probe.log
grep -v "^W1005\|OMP_NUM" gpurun_out/r05q/probe.log | tail -n 25
