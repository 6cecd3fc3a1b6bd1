#!/bin/bash
# Round-4 evidence: the bench lines, rocprofv3 kernel-trace statistics of the same commands, and the PMC passes (FETCH_SIZE / WRITE_SIZE /
# MFMA counters, each in its own run, never combined with a trace domain other than --kernel-trace).  Outputs under gpurun_out/ev4.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ev4
mkdir -p $O && rm -rf $O/pmc_*   # (a merge of two runs would average kernels of different builds)
cd /tmp && export TMPDIR=/tmp
# every bench line ends with its CPU baseline (10-15 s of 16 busy host threads); the training steps are ~270 launches per 15 ms and slow
# down by 10-15 % when the NEXT process starts on host cores that are still hot from it (measured: 15.5 -> 17.0-18.2 ms with identical
# kernel durations), so the launch-heavy workloads go first and a pause follows each baseline
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc_fetch.log 2>&1 && echo fetch-ok && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc_write.log 2>&1 && echo write-ok && \
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc_mfma.log 2>&1 && echo mfma-ok && \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_fe -- python3 $R/bench.py --workload frontend --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_fe.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_fe -- python3 $R/bench.py --workload frontend --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write_fe.log 2>&1 && echo fe-pmc-ok && \
ORCAI_HPS_VARIANTS=set3 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_h -- python3 $R/bench.py --workload hpsearch --steps 1 --warmup 1 --no-cpu-baseline --no-loss-curves > $O/pmc_fetch_h.log 2>&1 && \
ORCAI_HPS_VARIANTS=set3 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_h -- python3 $R/bench.py --workload hpsearch --steps 1 --warmup 1 --no-cpu-baseline --no-loss-curves > $O/pmc_write_h.log 2>&1 && echo hps-pmc-ok
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_t -- python3 $R/bench.py --workload train --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_t.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_t -- python3 $R/bench.py --workload train --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_write_t.log 2>&1 && echo train-pmc-ok
du -sh $O
