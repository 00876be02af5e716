#!/bin/bash
# GPU test suite + __graft_entry__.smoke(), as the driver runs them at round end (run through gpurun).
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 300 > gpurun_out/y_tests.log 2>&1; tail -5 gpurun_out/y_tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
