#!/bin/bash
# Round 4: issue / wait / LDS counters of the top predict kernels (with the fused block tail and with the two launches it replaces) and of the
# training step, one rocprofv3 --pmc pass per counter group (no tracing domains besides --kernel-trace).  Summaries: tools/summarize_pmc.py.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
PMCG=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA"
        "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_VMEM")
for what in ${1:-fused twolaunch train}; do
  i=0
  for grp in "${PMCG[@]}"; do
    i=$((i+1))
    d=$R/gpurun_out/pmc4_${what}_$i
    case $what in
      fused)     ORCAI_POOL_FUSED=12 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 $R/tools/debug_predict.py 300 128 > $d.log 2>&1 && echo $what-pass-$i-ok ;;
      twolaunch) ORCAI_POOL_FUSED=0  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 $R/tools/debug_predict.py 300 128 > $d.log 2>&1 && echo $what-pass-$i-ok ;;
      hps)       ORCAI_HPS_VARIANTS=set3 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --workload hpsearch --steps 1 --warmup 1 --no-cpu-baseline --no-loss-curves > $d.log 2>&1 && echo $what-pass-$i-ok ;;
      train)     rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 $R/tools/debug_train.py 64 > $d.log 2>&1 && echo $what-pass-$i-ok ;;
    esac
  done
  python3 $R/tools/summarize_pmc.py $R/gpurun_out/pmc4_${what}_1 $R/gpurun_out/pmc4_${what}_2 $R/gpurun_out/pmc4_${what}_3 > $R/gpurun_out/r04_pmc_issue_${what}.json
  rm -rf $R/gpurun_out/pmc4_${what}_[123]
done
