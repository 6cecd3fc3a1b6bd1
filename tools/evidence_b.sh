#!/bin/bash
# Round evidence, part B: rocprofv3 kernel-trace stats of the bench commands, then FETCH_SIZE / WRITE_SIZE in their own passes.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ev
mkdir -p $O
rm -rf $O/prof_* $O/pmc_*   # rocprofv3 names its files by PID: results of an earlier run would be summarised along with this one
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_predict -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_predict.log 2>&1 && echo predict-trace-ok && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python3 $R/bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_train.log 2>&1 && echo train-trace-ok && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_frontend -- python3 $R/bench.py --workload frontend --steps 20 --warmup 3 --no-cpu-baseline > $O/prof_frontend.log 2>&1 && echo frontend-trace-ok && \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1 && echo fetch-ok && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1 && echo write-ok && \
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_mfma.log 2>&1 && echo mfma-ok && \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_fe -- python3 $R/bench.py --workload frontend --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_fe.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_fe -- python3 $R/bench.py --workload frontend --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write_fe.log 2>&1 && echo fe-pmc-ok
find $O -name "*.csv" | head -40
