set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for n in 1 11; do
  rocprofv3 --hip-trace --stats -d $R/gpurun_out/hiptrace_$n -- python3 $R/tools/debug_graph_nodes.py $n > $R/gpurun_out/hiptrace_$n.log 2>&1
  f=$(ls $R/gpurun_out/hiptrace_$n/*/*hip_api_stats.csv | head -n 1)
  echo "== steps $n: $f"; grep -i "memset\|memcpy\|LaunchKernel" $f
done
