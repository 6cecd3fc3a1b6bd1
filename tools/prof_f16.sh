#!/bin/bash
# rocprofv3 kernel-trace stats of the f16 training step (hpsearch workload, one variant) and of the f16 inference forward.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_f16
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ORCAI_HPS_VARIANTS=${1:-set3} rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_f16 -- python3 $R/bench.py --workload hpsearch --steps 10 --warmup 3 --no-cpu-baseline --no-loss-curves > $O/train_f16.log 2>&1 && echo f16-ok
f=$(find $O/train_f16 -name "*kernel_stats.csv" | head -1); cp "$f" $O/train_f16_kernel_stats.csv
tail -c 1500 $O/train_f16.log
