#!/bin/bash
# round 5, final measurements: the driver-style bench lines (default 300 steps; and --steps 20 as the driver passes), the
# rocprofv3 kernel stats + PMC traffic of the headline configuration, the A-V and 256^3 and plain-DIA configurations,
# the launcher form of the bench on one rank, the slab-shape and rehearsal tables once more
set -o pipefail
out=gpurun_out/r05final; mkdir -p $out
timeout -k 10 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc=$?" | tee -a $out/summary.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $out/bench_steps20.json 2> $out/bench_steps20.err; echo "bench steps20 rc=$?" | tee -a $out/summary.log
timeout -k 10 600 bash tools/profile_bench.sh r05_final 512 dict cube > $out/prof_final.log 2>&1; echo "profile final rc=$?" | tee -a $out/summary.log
timeout -k 10 400 bash tools/profile_bench.sh r05_av 512 dict av 3 > $out/prof_av.log 2>&1; echo "profile av rc=$?" | tee -a $out/summary.log
timeout -k 10 400 bash tools/profile_bench.sh r05_256 256 dict cube > $out/prof_256.log 2>&1; echo "profile 256 rc=$?" | tee -a $out/summary.log
timeout -k 10 600 bash tools/profile_bench.sh r05_dia 512 dia cube > $out/prof_dia.log 2>&1; echo "profile dia rc=$?" | tee -a $out/summary.log
timeout -k 10 300 python bench.py --rehearse 4,8 --steps 300 --no-cpu-baseline > $out/bench_rehearse_4_8.json 2> $out/bench_rehearse_4_8.err; echo "bench rehearse 4,8 rc=$?" | tee -a $out/summary.log
timeout -k 10 300 python bench.py --rehearse 1,2 --steps 100 --no-cpu-baseline > $out/bench_rehearse_1_2.json 2> $out/bench_rehearse_1_2.err; echo "bench rehearse 1,2 rc=$?" | tee -a $out/summary.log
timeout -k 10 300 python bench.py --gpus 2 --devices 0,0 --steps 100 --no-cpu-baseline > $out/bench_2slabs.json 2> $out/bench_2slabs.err; echo "bench 2 slabs rc=$?" | tee -a $out/summary.log
ls gpurun_out/profiles_r05_* 2>/dev/null
cut -c1-400 $out/bench_default.json
