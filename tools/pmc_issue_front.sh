#!/bin/bash
# Issue/wait counters of the model kernels with block 1's front fused (orcai_block_front) on a 300 s recording, one pass per counter group.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export ORCAI_FUSE_FRONT=1
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmcb_$i -- python3 $R/tools/debug_predict.py 300 128 > $R/gpurun_out/pmcb_$i.log 2>&1 && echo pass-$i-ok
done
python3 $R/tools/summarize_pmc.py $R/gpurun_out/pmcb_1 $R/gpurun_out/pmcb_2 $R/gpurun_out/pmcb_3 > $R/gpurun_out/pmc_issue_front.json
