#!/bin/bash
# round 4, GPU call b: the structured form on runtime-shaped 2-D tiles -- parity first, then timings against linear tiles
out=gpurun_out/r04b; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_formats_dist.py tests/test_gpu_parity.py tests/test_vtk_output.py tests/test_gpu_host_program.py tests/test_gpu_default_policies.py -m gpu -q -x -rsx > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 15 $out/pytest.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
for wl in av3 hole lim av1; do
  EC3D_SAV_PATCH=0 timeout -k 10 200 python3 tools/ab_perf.py $wl linear >> $out/ab.log 2>> $out/ab.err
  timeout -k 10 200 python3 tools/ab_perf.py $wl patch >> $out/ab.log 2>> $out/ab.err
  EC3D_FUSE23=2 EC3D_FUSE51=2 timeout -k 10 200 python3 tools/ab_perf.py $wl patch_fused >> $out/ab.log 2>> $out/ab.err
done
done
cat $out/ab.log
