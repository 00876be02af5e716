#!/bin/bash
# Round 3, call AH: new eval tests + the eval / parity files, then bench extras
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_fullsize_golden.py tests/test_hip_fullsize_configs.py tests/test_hip_edges.py -x -q -m gpu --timeout 300 > $O/ah3_tests.log 2>&1; tail -5 $O/ah3_tests.log | cut -c1-300
