#!/bin/bash
# Round 3, call P: two-phase ("staged direction") owner-computes step: bit-identity with the one-phase step, then same-box A/B
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_hip_pull.py -x -q -m gpu --timeout 200 -k "two_phase" > $O/p3_tests.log 2>&1; tail -15 $O/p3_tests.log | cut -c1-300
for rep in 1 2; do
for dir in 0 1; do
  KGE_PULL_DIR=$dir timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/p3_tmp.json 2> $O/p3_tmp.err
  python - <<PY
import json
d=json.load(open('gpurun_out/p3_tmp.json'))
print("KGE_PULL_DIR=$dir", "ms_per_step %.4f" % d["ms_per_step"], "value %.3f G" % (d["value"]/1e9), "small", round(d["train_reference_default_batch"]["ms_per_step"]*1e3,2), "us")
PY
done; done | tee $O/p3_ab.log
tail -3 $O/p3_tmp.err
KGE_PULL_DIR=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/p3_p -o t -- python bench.py --no-cpu-baseline --no-extra-configs > $O/p3_prof.log 2>&1
python tools/rocpd_summary.py $O/p3_p/t_results.db $O/p3_kernels.md > /dev/null; grep "k_pull" $O/p3_kernels.md | cut -c1-200 | head -8
rm -rf $O/p3_p
