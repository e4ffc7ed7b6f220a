# Experiment (GPU box): A-V bench with several SpMV workgroup counts, then a profile of one of them.
set -e
for nb in 0 1656 2208 3312 4416; do
  if [ $nb = 0 ]; then unset EC3D_NBLK_SPMV; else export EC3D_NBLK_SPMV=$nb; fi
  python bench.py --workload av --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/av_nb$nb.json 2> gpurun_out/av_nb$nb.err
done
export EC3D_NBLK_SPMV=2208
bash tools/profile_bench.sh r02_av_nb2208 512 dict av > gpurun_out/prof_av_nb2208.log 2>&1
