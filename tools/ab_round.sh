#!/bin/bash
# One gpurun call for an A/B: the parity gate first (a variant that breaks results is not worth timing), then the variants
# back to back on the same box (boxes differ by up to 10 %).  Edit the loop; EC3D_LIB=<other build> selects another library.
set -e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_formats_dist.py -x -q > gpurun_out/ab_pytest.log 2>&1 || { tail -20 gpurun_out/ab_pytest.log; exit 1; }
tail -2 gpurun_out/ab_pytest.log
for rep in 1 2; do
  for wl in cube256 av3; do
    EC3D_KEEP=0 python3 tools/ab_perf.py $wl all_nontemporal
    python3 tools/ab_perf.py $wl policy
  done
done
