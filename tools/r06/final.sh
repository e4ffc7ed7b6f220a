#!/bin/bash
# round 6, final measurements: the driver-style bench lines (default 300 steps; and --steps 20 --warmup 5 as the driver passes),
# rocprofv3 kernel stats + PMC traffic of the headline configuration, of BASELINE config 3 at 256^3 (av256), the 21 M A-V system,
# 256^3 and plain DIA; the multi-GPU forms of the bench on one card; the av256 test with its printed numbers
set -o pipefail
out=gpurun_out/r06final; mkdir -p $out
timeout -k 10 500 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc=$?" | tee -a $out/summary.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $out/bench_steps20.json 2> $out/bench_steps20.err; echo "bench steps20 rc=$?" | tee -a $out/summary.log
timeout -k 10 600 bash tools/profile_bench.sh r06_final 512 dict cube > $out/prof_final.log 2>&1; echo "profile final rc=$?" | tee -a $out/summary.log
timeout -k 10 500 bash tools/profile_bench.sh r06_av256 256 dict av256 > $out/prof_av256.log 2>&1; echo "profile av256 rc=$?" | tee -a $out/summary.log
timeout -k 10 400 bash tools/profile_bench.sh r06_av 512 dict av 3 > $out/prof_av.log 2>&1; echo "profile av rc=$?" | tee -a $out/summary.log
timeout -k 10 400 bash tools/profile_bench.sh r06_256 256 dict cube > $out/prof_256.log 2>&1; echo "profile 256 rc=$?" | tee -a $out/summary.log
timeout -k 10 600 bash tools/profile_bench.sh r06_dia 512 dia cube > $out/prof_dia.log 2>&1; echo "profile dia rc=$?" | tee -a $out/summary.log
timeout -k 10 300 python bench.py --workload av256 --steps 200 --no-cpu-baseline > $out/bench_av256.json 2> $out/bench_av256.err; echo "bench av256 rc=$?" | tee -a $out/summary.log
timeout -k 10 300 python bench.py --rehearse 4,8 --steps 300 --no-cpu-baseline > $out/bench_rehearse_4_8.json 2> $out/bench_rehearse_4_8.err; echo "bench rehearse 4,8 rc=$?" | tee -a $out/summary.log
timeout -k 10 400 python bench.py --gpus 2 --devices 0,0 --steps 100 --no-cpu-baseline > $out/bench_2slabs.json 2> $out/bench_2slabs.err; echo "bench 2 slabs rc=$?" | tee -a $out/summary.log
timeout -k 10 400 python bench.py --gpus 1 --devices 0 --steps 100 --no-cpu-baseline > $out/bench_1rank_plain_form.json 2> $out/bench_1rank_plain_form.err; echo "bench plain form 1 rank rc=$?" | tee -a $out/summary.log
timeout -k 10 600 python -m pytest tests/test_gpu_av256.py -q -m gpu -s > $out/test_av256.log 2>&1; echo "test av256 rc=$?" | tee -a $out/summary.log
grep -v "^\.\|^$" $out/test_av256.log | cut -c1-400 | tail -n 20
ls gpurun_out/profiles_r06_* 2>/dev/null
cut -c1-600 $out/bench_default.json
