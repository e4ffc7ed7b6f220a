#!/bin/bash
# round 4, call t: the whole GPU suite, smoke, the default bench line, and the RCCL path with one rank (--force-dist)
out=gpurun_out/r04t; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -rsx > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 4 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $out/smoke.log
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-300 $out/bench.json
timeout -k 10 300 python bench.py --force-dist --steps 10 --warmup 3 > $out/bench_force_dist.json 2> $out/bench_force_dist.err; echo "force-dist rc=$?"; cut -c1-300 $out/bench_force_dist.json
