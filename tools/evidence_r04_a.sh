#!/bin/bash
# Round-4 evidence: the bench lines, rocprofv3 kernel-trace statistics of the same commands, and the PMC passes (FETCH_SIZE / WRITE_SIZE /
# MFMA counters, each in its own run, never combined with a trace domain other than --kernel-trace).  Outputs under gpurun_out/ev4.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ev4
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# every bench line ends with its CPU baseline (10-15 s of 16 busy host threads); the training steps are ~270 launches per 15 ms and slow
# down by 10-15 % when the NEXT process starts on host cores that are still hot from it (measured: 15.5 -> 17.0-18.2 ms with identical
# kernel durations), so the launch-heavy workloads go first and a pause follows each baseline
python3 $R/bench.py --workload train > $O/bench_train.json 2> $O/bench_train.err && echo bench-train-ok && sleep 20 && \
python3 $R/bench.py --workload hpsearch --steps 10 --warmup 3 > $O/bench_hpsearch.json 2> $O/bench_hpsearch.err && echo bench-hpsearch-ok && sleep 20 && \
python3 $R/bench.py > $O/bench_predict.json 2> $O/bench_predict.err && echo bench-predict-ok && sleep 20 && \
python3 $R/bench.py --workload frontend > $O/bench_frontend.json 2> $O/bench_frontend.err && echo bench-frontend-ok && sleep 20 && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_predict -- python3 $R/bench.py > $O/prof_predict.log 2>&1 && echo predict-trace-ok && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python3 $R/bench.py --workload train --no-cpu-baseline > $O/prof_train.log 2>&1 && echo train-trace-ok && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_frontend -- python3 $R/bench.py --workload frontend --no-cpu-baseline > $O/prof_frontend.log 2>&1 && echo frontend-trace-ok && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hpsearch -- python3 $R/bench.py --workload hpsearch --steps 10 --warmup 3 --no-cpu-baseline --no-loss-curves > $O/prof_hpsearch.log 2>&1 && echo hpsearch-trace-ok
for d in predict train frontend hpsearch; do f=$(find $O/prof_$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${d}_kernel_stats.csv; done
# the raw kernel traces are large: keep the statistics and the counter collections only
find $O -name "*kernel_trace.csv" -path "*prof_*" -delete
du -sh $O
