#!/bin/bash
# round 5, call ae: whole GPU suite with plan 5 as the default of small five-launch slabs (no -x: list what depends on the old default)
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/r05_ae_suite.log 2>&1
rc=$?; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r05_ae_suite.log | tail -n 30; exit 0
