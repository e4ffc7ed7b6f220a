#!/bin/bash
# round 6, call k: the whole GPU suite on the round's build (no -x: every failure in one call)
out=$(pwd)/gpurun_out/r06k; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -q -m gpu -rs > $out/pytest.log 2>&1
rc=$?; grep -n "^FAILED\|^ERROR\|passed\|failed" $out/pytest.log | tail -n 25; exit $rc
