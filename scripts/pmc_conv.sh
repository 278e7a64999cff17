#!/bin/bash
# PMC passes over one conv shape (run on the GPU box): scripts/pmc_conv.sh <shape> <outdir>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
shape=$1; out=$2
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $set --output-format csv -d $out/p$i -o p -- python3 scripts/conv_bench.py $shape > $out.p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out.p$i.log; exit 1; }
done
echo done
