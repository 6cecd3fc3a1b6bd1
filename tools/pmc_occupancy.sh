#!/bin/bash
# Occupancy / SIMD-busy counters of the block-1 kernels at the benchmark's launch size (1 200 s of audio: chunks of 122 snippets), fused tail vs two launches.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for what in 8 0; do
  d=$R/gpurun_out/pmc_occ_$what
  ORCAI_POOL_FUSED=$what rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $d -- python3 $R/tools/debug_predict.py 1200 128 > $d.log 2>&1 && echo occ-$what-ok
  python3 $R/tools/summarize_pmc.py $d > $R/gpurun_out/r04_pmc_occupancy_fused$what.json
  rm -rf $d
done
