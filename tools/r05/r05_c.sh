#!/bin/bash
# round 5, call c: the RCCL rank handle (one-rank job, rehearsals), slab plans again, bench forms
set -o pipefail
mkdir -p gpurun_out/r05c
python -m pytest tests/test_gpu_rccl_rank.py -x -q -s > gpurun_out/r05c/rccl.log 2>&1; echo "rccl rc=$?" | tee -a gpurun_out/r05c/summary.log
python -m pytest tests/test_gpu_slab_plans.py -x -q > gpurun_out/r05c/slab_plans.log 2>&1; echo "slab_plans rc=$?" | tee -a gpurun_out/r05c/summary.log
timeout -k 10 300 python tools/rank_rehearsal.py 200 > gpurun_out/r05c/rank_rehearsal.log 2>&1; echo "rehearsal rc=$?" | tee -a gpurun_out/r05c/summary.log
timeout -k 10 300 python bench.py --gpus 2 --devices 0,0 --steps 50 --no-cpu-baseline > gpurun_out/r05c/bench_2slabs.json 2> gpurun_out/r05c/bench_2slabs.err; echo "bench 2 slabs rc=$?" | tee -a gpurun_out/r05c/summary.log
timeout -k 10 300 python bench.py --force-dist --steps 50 --no-cpu-baseline > gpurun_out/r05c/bench_force_dist.json 2> gpurun_out/r05c/bench_force_dist.err; echo "bench force-dist rc=$?" | tee -a gpurun_out/r05c/summary.log
timeout -k 10 300 python bench.py --rehearse 3,8 --steps 200 --no-cpu-baseline > gpurun_out/r05c/bench_rehearse_3_8.json 2> gpurun_out/r05c/bench_rehearse_3_8.err; echo "bench rehearse rc=$?" | tee -a gpurun_out/r05c/summary.log
tail -n 8 gpurun_out/r05c/rccl.log; tail -n 3 gpurun_out/r05c/slab_plans.log; cat gpurun_out/r05c/rank_rehearsal.log; tail -n 3 gpurun_out/r05c/*.err
