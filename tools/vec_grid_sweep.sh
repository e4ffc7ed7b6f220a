#!/bin/bash
# Experiment (GPU box): workgroup counts of the vector kernels K2 / K4 / K5, each on a grid of its own.
# Usage: tools/vec_grid_sweep.sh <grid> ; prints ms per iteration and per kernel for every combination.
GRID=${1:-512}
mkdir -p gpurun_out
for k4 in 256 512 768; do for k2 in 512 768; do for k5 in 512 768; do
  EC3D_NBLK_K4=$k4 EC3D_NBLK_K2=$k2 EC3D_NBLK_K5=$k5 python3 bench.py --grid $GRID --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null \
   | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('k4=$k4 k2=$k2 k5=$k5', round(d['ms_per_step'],4), {k:round(v['ms'],4) for k,v in d['kernels'].items()})"
done; done; done
