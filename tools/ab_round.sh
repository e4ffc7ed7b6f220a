#!/bin/bash
set -e
for t in 0 6168 12336 24 6144 12288 16416 48; do EC3D_TAIL=$t python3 tools/ab_perf.py cube512 tail$t; done
