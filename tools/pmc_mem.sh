#!/bin/bash
# Memory-pipeline counters (TA / TCP / TCC) of the model kernels on a 300 s recording; at most 4 counters of a block per pass
# (more than that: "Request exceeds the capabilities of the hardware to collect"), every pass under its own timeout.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
i=0
for grp in "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum" "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmcm_$i -- python3 $R/tools/debug_predict.py 300 128 > $R/gpurun_out/pmcm_$i.log 2>&1 && echo pass-$i-ok || { echo pass-$i-failed; grep -m2 -i "exceeds\|error" $R/gpurun_out/pmcm_$i.log; }
done
