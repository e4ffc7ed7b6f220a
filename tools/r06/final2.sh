#!/bin/bash
# round 6, second session, final measurements of the 512^3 configuration (the only one whose launches changed): driver-style
# bench lines, rocprofv3 kernel stats + PMC traffic, the 8-rank rehearsal and the one-card multi-GPU forms
set -o pipefail
out=gpurun_out/r06final2; mkdir -p $out
timeout -k 10 500 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc=$?" | tee -a $out/summary.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $out/bench_steps20.json 2> $out/bench_steps20.err; echo "bench steps20 rc=$?" | tee -a $out/summary.log
timeout -k 10 700 bash tools/profile_bench.sh r06_final 512 dict cube > $out/prof_final.log 2>&1; echo "profile final rc=$?" | tee -a $out/summary.log
timeout -k 10 300 python bench.py --rehearse 4,8 --steps 300 --no-cpu-baseline > $out/bench_rehearse_4_8.json 2> $out/bench_rehearse_4_8.err; echo "bench rehearse 4,8 rc=$?" | tee -a $out/summary.log
timeout -k 10 400 python bench.py --gpus 2 --devices 0,0 --steps 100 --no-cpu-baseline > $out/bench_2slabs.json 2> $out/bench_2slabs.err; echo "bench 2 slabs rc=$?" | tee -a $out/summary.log
ls gpurun_out/profiles_r06_* 2>/dev/null
python - <<P
import json
for f in ("bench_default", "bench_steps20"):
    d = json.load(open("$out/" + f + ".json"))
    print(f, "ms_per_step", round(d["ms_per_step"], 4), "value", round(d["value"] / 1e9, 2), "G; roofline frac", round(d["roofline"]["frac"], 3),
          {k: (round(v["ms"] * 1e3, 1), v["bytes_per_row"], round(v["GBps"] / 8000, 3)) for k, v in d["kernels"].items()},
          "iter_hbm_frac", round(d["iter_hbm_frac"], 3), d["config"].get("vector_placement", {}).get("candidate_us_per_iteration"),
          {k: round(d[k]["value"] / 1e9, 2) for k in ("av", "cube256", "av256") if k in d and "value" in d[k]},
          "cpu", d.get("cpu_baseline", {}).get("value"))
P
