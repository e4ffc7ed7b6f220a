#!/bin/bash
# round 6, call n: 2-D tiles with the plane above requested a step ahead -- bitwise tests, then the 512^3 line
out=$(pwd)/gpurun_out/r06n; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slab_plans.py tests/test_gpu_config4.py tests/test_gpu_default_policies.py -q -m gpu -x > $out/pytest.log 2>&1
rc=$?; tail -n 5 $out/pytest.log | cut -c1-300; [ $rc -ne 0 ] && exit 1
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-side-workloads --no-spmv-dia > $out/bench_$i.json 2> $out/bench_$i.err
python - <<P
import json
d=json.load(open("$out/bench_$i.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], {k:(round(v["ms"]*1e3,1), round(v["GBps"]/8000,3)) for k,v in d["kernels"].items()}, "spmv", d["spmv"]["ms"])
P
done
timeout -k 10 300 python bench.py --grid 256 --no-cpu-baseline --no-side-workloads --no-spmv-dia > $out/bench_256.json 2> $out/bench_256.err
python - <<P
import json
d=json.load(open("$out/bench_256.json"))
print("256: ms_per_step", d["ms_per_step"], "value", d["value"], {k:(round(v["ms"]*1e3,1), round(v["GBps"]/8000,3)) for k,v in d["kernels"].items()}, "spmv", d["spmv"]["ms"])
P
