#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/gaps
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/predict -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $O/predict.log 2>&1 && echo predict-ok
python3 $R/tools/gap_analysis.py $O/predict --last-frac 0.5 --top 40 > $O/predict.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
cat $O/predict.txt
