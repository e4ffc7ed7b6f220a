#!/bin/bash
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/r05_aj_suite.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r05_aj_suite.log | tail -n 20
