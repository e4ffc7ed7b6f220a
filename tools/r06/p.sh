#!/bin/bash
# round 6, call p: tools/zmarch_bench2.hip -- K2-in-K3's traffic (16 B read + 8 B written per row) under the z-march's structure
out=$(pwd)/gpurun_out/r06p; mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/zmarch_bench2.hip -o /tmp/zb2 2> $out/build.err || { tail $out/build.err; exit 1; }
timeout -k 10 240 /tmp/zb2 > $out/zb2.log 2>&1
rc=$?; cat $out/zb2.log; exit $rc
