#!/bin/bash
# kernel-trace stats of the learn step (net5, batch 128, fp32) with the stream-K GEMM and with its one-workgroup-per-tile twin
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03_learn_sk -- python3 tools/learn_bench.py 10 128 > $O/r03_learn_sk_line.json 2>> $O/prof_r03_learn.err || exit 1
cp "$(find $O/prof_r03_learn_sk -name '*kernel_stats.csv' | head -1)" $O/r03_learn_sk_kernel_stats.csv
export TZ_LEARN_GEMM=tile
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03_learn_tile -- python3 tools/learn_bench.py 10 128 > $O/r03_learn_tile_line.json 2>> $O/prof_r03_learn.err || exit 1
cp "$(find $O/prof_r03_learn_tile -name '*kernel_stats.csv' | head -1)" $O/r03_learn_tile_kernel_stats.csv
unset TZ_LEARN_GEMM
cut -c1-150 $O/r03_learn_sk_kernel_stats.csv | head -12; cut -c1-150 $O/r03_learn_tile_kernel_stats.csv | head -10
