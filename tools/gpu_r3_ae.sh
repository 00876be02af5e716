#!/bin/bash
# Round 3, call AE: where k_eval_gemm's waves spend their time (SQ counters, one pass, --kernel-trace only), C3 and C2 shapes
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
for sh in c3 c2; do
SHAPE=$sh timeout 200 python tools/eval_only.py 2>&1 | tail -1
SHAPE=$sh REPS=1 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS --kernel-trace -d $O/ae_pmc_$sh -o ev -- python tools/eval_only.py > $O/ae_pmc_$sh.log 2>&1
SHAPE=$sh REPS=1 timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace -d $O/ae_pmc2_$sh -o ev -- python tools/eval_only.py > $O/ae_pmc2_$sh.log 2>&1
python tools/rocpd_pmc.py $O/ae3_pmc_gemm_$sh.json "SQ counters of the $sh eval pass (tools/eval_only.py), two passes" $(find $O/ae_pmc_$sh $O/ae_pmc2_$sh -name "*.db")
rm -rf $O/ae_pmc_$sh $O/ae_pmc2_$sh
done
python - <<'PY'
import json
for sh in ("c3","c2"):
    d=json.load(open(f"gpurun_out/ae3_pmc_gemm_{sh}.json"))["kernels"]
    for k,v in d.items():
        if "k_eval_gemm" in k:
            print(sh,k)
            for c,x in v.items(): print("   %-32s %14.0f  (%.0f us)" % (c, x["avg"], x["avg_duration_us_in_this_pass"]))
PY
