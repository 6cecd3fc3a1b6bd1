#!/bin/bash
# The full GPU suite plus smoke() as one box command; exits non-zero when either fails.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu.ids | cut -c1-1500 > gpurun_out/pytest_full.log
rc_pytest=${PIPESTATUS[0]}
echo "pytest rc $rc_pytest"; tail -n 5 gpurun_out/pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
rc_smoke=$?
echo "smoke rc $rc_smoke"; tail -n 3 gpurun_out/smoke.log
exit $(( rc_pytest || rc_smoke ))
