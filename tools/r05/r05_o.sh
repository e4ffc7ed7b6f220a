#!/bin/bash
# round 5, call o: the A-V tests of the RCCL rank handle; one rank of BASELINE config 5 on 8 and 4 GPUs (and config 3 on 2) alone
# on one card through the RCCL driver, beside the undivided system through the same driver (one-rank job)
set -o pipefail
out=gpurun_out/r05o; mkdir -p $out
python -m pytest tests/test_gpu_rccl_rank.py -x -q > $out/rccl.log 2>&1; echo "rccl tests rc=$?" | tee -a $out/summary.log
REHEARSE_AV=1 timeout -k 10 500 python tools/rank_rehearsal.py 200 > $out/av_rehearsal.log 2>&1; echo "av rehearsal rc=$?" | tee -a $out/summary.log
for n in 4 8; do
  devs=$(python -c "print(','.join(['0']*$n))")
  timeout -k 10 300 python bench.py --gpus $n --devices $devs --steps 50 --no-cpu-baseline > $out/bench_${n}slabs.json 2> $out/bench_${n}slabs.err; echo "bench $n slabs rc=$?" | tee -a $out/summary.log
done
tail -n 3 $out/rccl.log; grep -v "version\|Hostname\|Librccl\|amdgpu.ids" $out/av_rehearsal.log
