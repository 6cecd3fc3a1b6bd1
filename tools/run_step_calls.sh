#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/calls
python3 $R/tools/step_calls.py f32 > $R/gpurun_out/calls/f32.txt 2>&1 && python3 $R/tools/step_calls.py f16 set3 > $R/gpurun_out/calls/f16_set3.txt 2>&1 && python3 $R/tools/step_calls.py f16 set1 > $R/gpurun_out/calls/f16_set1.txt 2>&1
tail -3 $R/gpurun_out/calls/*.txt
