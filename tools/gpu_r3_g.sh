#!/bin/bash
# Round 3, call G: own-step vs atomic step over shapes; full GPU suite; bench line
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 500 python tools/own_perf.py > $O/g3_own_perf.md 2>&1; cat $O/g3_own_perf.md
timeout 600 python -m pytest tests -q -m gpu --timeout 300 > $O/g3_tests.log 2>&1; tail -4 $O/g3_tests.log | cut -c1-300
