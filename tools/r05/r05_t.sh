#!/bin/bash
# round 5, call t: what one solve costs around its iterations at the reference's shipped size
timeout -k 10 300 python3 tools/solve_overhead.py 15 > gpurun_out/r05_t.log 2>&1; rc=$?
cat gpurun_out/r05_t.log; exit $rc
