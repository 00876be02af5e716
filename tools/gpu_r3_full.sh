#!/bin/bash
# Round 3: full GPU suite (+ smoke + default bench line)
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --timeout 300 > $O/full_tests.log 2>&1; tail -5 $O/full_tests.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/full_smoke.log 2>&1; tail -2 $O/full_smoke.log
