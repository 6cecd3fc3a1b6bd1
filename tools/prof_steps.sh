#!/bin/bash
# rocprofv3 kernel-trace stats of the training step, f32 (train workload) and f16 (hpsearch workload, set3 only): where a step's time goes.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_steps
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_f32 -- python3 $R/bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > $O/train_f32.log 2>&1 && echo f32-ok && \
ORCAI_HPS_VARIANTS=set3 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_f16 -- python3 $R/bench.py --workload hpsearch --steps 10 --warmup 3 --no-cpu-baseline --no-loss-curves > $O/train_f16.log 2>&1 && echo f16-ok
for d in train_f32 train_f16; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp "$f" $O/${d}_kernel_stats.csv; done
ls $O
