#!/bin/bash
# kernel trace of 20 tz_net_eval calls at batch 128 (and 1): what a call's 0.5 ms is made of
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for B in 128 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03_agent_$B -- python3 tools/agent_trace.py $B > /dev/null 2>> $O/prof_r03_agent.err || exit 1
  cp "$(find $O/prof_r03_agent_$B -name '*kernel_stats.csv' | head -1)" $O/r03_agent_b${B}_kernel_stats.csv
  echo "== batch $B"; cut -c1-170 $O/r03_agent_b${B}_kernel_stats.csv | head -9
done
