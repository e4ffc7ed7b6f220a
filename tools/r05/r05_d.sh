#!/bin/bash
# round 5, call d: plan 4 (split fused producers), band-placement parking, bench contract, rehearsals with and without the split
set -o pipefail
mkdir -p gpurun_out/r05d
python -m pytest tests/test_gpu_slab_plans.py tests/test_gpu_rccl_rank.py -x -q > gpurun_out/r05d/slab_plans.log 2>&1; echo "slab_plans+rccl rc=$?" | tee -a gpurun_out/r05d/summary.log
python -m pytest tests/test_gpu_config4.py -x -q -s -k "slab_shapes" > gpurun_out/r05d/config4.log 2>&1; echo "config4 slab shapes rc=$?" | tee -a gpurun_out/r05d/summary.log
python -m pytest tests/test_gpu_edge_cases.py tests/test_bench_contract.py tests/test_gpu_multi.py -x -q > gpurun_out/r05d/misc.log 2>&1; echo "edge+contract+multi rc=$?" | tee -a gpurun_out/r05d/summary.log
timeout -k 10 300 python tools/rank_rehearsal.py 200 > gpurun_out/r05d/rank_rehearsal_split.log 2>&1; echo "rehearsal rc=$?" | tee -a gpurun_out/r05d/summary.log
EC3D_SLAB_FSPLIT=0 timeout -k 10 300 python tools/rank_rehearsal.py 200 > gpurun_out/r05d/rank_rehearsal_nosplit.log 2>&1; echo "rehearsal nosplit rc=$?" | tee -a gpurun_out/r05d/summary.log
timeout -k 10 300 python tools/slab_shapes.py 100 > gpurun_out/r05d/slab_shapes.log 2>&1; echo "slab_shapes rc=$?" | tee -a gpurun_out/r05d/summary.log
tail -n 4 gpurun_out/r05d/slab_plans.log gpurun_out/r05d/config4.log gpurun_out/r05d/misc.log; grep -h "512x512x512" gpurun_out/r05d/rank_rehearsal_split.log gpurun_out/r05d/rank_rehearsal_nosplit.log; cat gpurun_out/r05d/slab_shapes.log
