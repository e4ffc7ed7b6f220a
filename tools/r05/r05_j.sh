#!/bin/bash
# round 5, call j: the whole GPU suite as the driver runs it; the counters rocprofv3 offers on this box (for the A-V SpMV question)
set -o pipefail
mkdir -p gpurun_out/r05j
( time python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r05j/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?" | tee -a gpurun_out/r05j/summary.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $GRAFT_REPO_ROOT/gpurun_out/r05j/list_avail.txt 2>&1; echo "list-avail rc=$?" | tee -a $GRAFT_REPO_ROOT/gpurun_out/r05j/summary.log
cd $GRAFT_REPO_ROOT
tail -n 6 gpurun_out/r05j/pytest_gpu.log
grep -i -c "TCC" gpurun_out/r05j/list_avail.txt
