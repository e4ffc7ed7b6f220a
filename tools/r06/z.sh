#!/bin/bash
# round 6, call z: X groups as launches of their own on the iteration's stream (default of the undivided three-launch handle):
# bitwise tests, then the headline with EC3D_XASYNC=0 and by default, alternating
out=$(pwd)/gpurun_out/r06z; mkdir -p $out
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config4.py tests/test_gpu_edge_cases.py tests/test_gpu_default_policies.py tests/test_gpu_slab_plans.py tests/test_gpu_rank_loopback.py -q -m gpu -x > $out/pytest.log 2>&1
rc=$?; tail -n 4 $out/pytest.log | cut -c1-300; [ $rc -ne 0 ] && { grep -n "^E " $out/pytest.log | head -20; exit 1; }
for i in 1 2 3; do
  EC3D_XASYNC=0 timeout -k 10 300 python3 tools/ab_perf.py cube512 K4_applies_the_group 2>> $out/ab.err | tee -a $out/ab.log || exit 1
  timeout -k 10 300 python3 tools/ab_perf.py cube512 group_launch_same_stream 2>> $out/ab.err | tee -a $out/ab.log || exit 1
done
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-side-workloads --no-spmv-dia > $out/bench_20.json 2> $out/bench_20.err || { tail $out/bench_20.err; exit 1; }
python - <<P
import json
d=json.load(open("$out/bench_20.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", {k:d["roofline"][k] for k in ("kernel","frac","avg_launch_ms")}, {k:(round(v["ms"]*1e3,1), v["bytes_per_row"], round(v["GBps"]/8000,3)) for k,v in d["kernels"].items()}, d["config"].get("vector_placement"))
P
