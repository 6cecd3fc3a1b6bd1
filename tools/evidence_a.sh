#!/bin/bash
# Round evidence, part A: GPU parity suite + the three bench lines.  Run on the GPU box: bash tools/evidence_a.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ev
mkdir -p $O
cd $R
python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1 && tail -3 $O/pytest_gpu.log && \
python bench.py > $O/bench_predict.json 2> $O/bench_predict.err && tail -c 600 $O/bench_predict.json && \
python bench.py --workload frontend > $O/bench_frontend.json 2> $O/bench_frontend.err && tail -c 300 $O/bench_frontend.json && \
python bench.py --workload train > $O/bench_train.json 2> $O/bench_train.err && tail -c 600 $O/bench_train.json
