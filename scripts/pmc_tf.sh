#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $out/p$i -o p -- python3 scripts/tf_bench.py > $out.p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out.p$i.log; exit 1; }
done
echo done
