#!/bin/bash
# round 6, call k2: the whole -m gpu suite on the tree with the new request modes, the placement search and the X-group launches
out=$(pwd)/gpurun_out/r06k2; mkdir -p $out
timeout -k 10 1150 python -m pytest tests -q -m gpu -x --durations=15 > $out/pytest.log 2>&1
rc=$?; tail -n 25 $out/pytest.log | cut -c1-200; exit $rc
