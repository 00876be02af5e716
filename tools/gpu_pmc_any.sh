#!/bin/bash
# Counter passes (one set per pass, --kernel-trace only) of an arbitrary command; per (kernel, grid) JSON.
# Usage (through gpurun): TAG=x ENVS="ONLY=..." bash tools/gpu_pmc_any.sh <command ...>
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
TAG=${TAG:-pmc}
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_IFETCH" \
           "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  env $ENVS timeout 300 rocprofv3 --pmc $set --kernel-trace -d $O/_q$i -o b -- "$@" > $O/${TAG}_q$i.log 2>&1 || echo "pass $i failed: $set"
done
python tools/rocpd_pmc.py $O/${TAG}_pmc.json "$*; one counter set per pass" $(find $O/_q* -name '*.db') | tail -1
rm -rf $O/_q*
