#!/bin/bash
# round 5, call m: a scalar-only rendezvous inside a kernel against a kernel boundary (the building block of a three-launch
# iteration for the reference's shipped problem sizes)
mkdir -p gpurun_out/r05m
hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_bench.hip -o /tmp/gbb && timeout -k 10 120 /tmp/gbb | tee gpurun_out/r05m/grid_barrier.log
timeout -k 10 120 /tmp/gbb | tee -a gpurun_out/r05m/grid_barrier.log
