#!/bin/bash
# round 6, call t: the adopted patch_pair request modes (default build): bitwise tests incl. the 512^3 fixtures, then a bench line
out=$(pwd)/gpurun_out/r06t; mkdir -p $out
timeout -k 10 1000 python -m pytest tests/test_gpu_formats_dist.py tests/test_gpu_fullsize.py tests/test_gpu_config4.py tests/test_gpu_multi.py tests/test_gpu_timeloop.py -q -m gpu -x > $out/pytest.log 2>&1
rc=$?; tail -n 5 $out/pytest.log | cut -c1-300; [ $rc -ne 0 ] && exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-side-workloads --no-spmv-dia > $out/bench_20.json 2> $out/bench_20.err || { tail $out/bench_20.err; exit 1; }
python - <<P
import json
d=json.load(open("$out/bench_20.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"], {k:(round(v["ms"]*1e3,1), round(v["GBps"]/8000,3)) for k,v in d["kernels"].items()})
P
# do the vectors' physical pages matter?  five handles alive together in one process
timeout -k 10 400 python tools/vec_place_probe.py 5 > $out/vec_place.log 2>&1; cat $out/vec_place.log | tail -20
