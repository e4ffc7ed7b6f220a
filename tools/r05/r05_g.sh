#!/bin/bash
# round 5, call g: partial sums collapsed by the producers' last workgroup (publish_partials): parity suites, then the rehearsal A/B
set -o pipefail
mkdir -p gpurun_out/r05g
python -m pytest tests/test_gpu_slab_plans.py tests/test_gpu_rccl_rank.py tests/test_gpu_multi.py tests/test_gpu_formats_dist.py tests/test_gpu_timeloop.py tests/test_gpu_two_process.py -x -q > gpurun_out/r05g/multi.log 2>&1; echo "multi suites rc=$?" | tee -a gpurun_out/r05g/summary.log
python -m pytest tests/test_gpu_config4.py -x -q -k "eight_slabs or twenty or slab_shapes" > gpurun_out/r05g/config4.log 2>&1; echo "config4 rc=$?" | tee -a gpurun_out/r05g/summary.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_default_policies.py tests/test_gpu_host_program.py -x -q > gpurun_out/r05g/single.log 2>&1; echo "single-GPU suites rc=$?" | tee -a gpurun_out/r05g/summary.log
for fold in 1 0 1 0; do
  echo "== EC3D_FOLD=$fold" >> gpurun_out/r05g/fold_ab.log
  EC3D_FOLD=$fold REHEARSE_ONLY="512,512,8,4;512,512,4,2;256,256,8,3;64,128,8,3" timeout -k 10 200 python tools/rank_rehearsal.py 300 >> gpurun_out/r05g/fold_ab.log 2>&1
done
tail -n 3 gpurun_out/r05g/multi.log gpurun_out/r05g/config4.log gpurun_out/r05g/single.log
grep -v "version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r05g/fold_ab.log
