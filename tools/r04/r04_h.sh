#!/bin/bash
# round 4 profiles: the bench line, then rocprofv3 kernel trace + PMC passes of the four configurations
out=gpurun_out/r04h; mkdir -p $out
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-300 $out/bench.json
timeout -k 10 500 bash tools/profile_bench.sh r04_final 512 dict cube > $out/prof_final.log 2>&1; echo "prof final rc=$?"
timeout -k 10 400 bash tools/profile_bench.sh r04_av 512 dict av 3 > $out/prof_av.log 2>&1; echo "prof av rc=$?"
timeout -k 10 400 bash tools/profile_bench.sh r04_256 256 dict cube > $out/prof_256.log 2>&1; echo "prof 256 rc=$?"
timeout -k 10 500 bash tools/profile_bench.sh r04_dia 512 dia cube > $out/prof_dia.log 2>&1; echo "prof dia rc=$?"
ls gpurun_out/profiles_r04_* 2>/dev/null
