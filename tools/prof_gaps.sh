#!/bin/bash
# Kernel traces of the three workloads for tools/gap_analysis.py (idle time between kernels).  Outputs: gpurun_out/gaps/*.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/gaps
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/hps -- python3 $R/bench.py --workload hpsearch --steps 10 --warmup 3 --no-cpu-baseline --no-loss-curves > $O/hps.log 2>&1 && echo hps-ok && \
rocprofv3 --kernel-trace --output-format csv -d $O/train -- python3 $R/bench.py --workload train --steps 20 --warmup 3 --no-cpu-baseline > $O/train.log 2>&1 && echo train-ok && \
rocprofv3 --kernel-trace --output-format csv -d $O/predict -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $O/predict.log 2>&1 && echo predict-ok
for w in hps train predict; do python3 $R/tools/gap_analysis.py $O/$w --last-frac 0.4 > $O/$w.txt 2>&1; done
find $O -name "*kernel_trace.csv" -delete
cat $O/hps.txt | head -40
