#!/bin/bash
out=gpurun_out/r04g3; mkdir -p $out
for mode in overlap nofiles sync; do
  timeout -k 10 400 python3 tools/config_runs.py --mode $mode >> $out/config_runs.log 2>> $out/config_runs.err || { echo "mode $mode failed"; tail -5 $out/config_runs.err; }
  tail -n 2 $out/config_runs.log
done
python -m pytest tests/test_gpu_host_program.py tests/test_vtk_output.py tests/test_gpu_timeloop.py tests/test_gpu_multi.py -m gpu -q 2>&1 | tail -3
