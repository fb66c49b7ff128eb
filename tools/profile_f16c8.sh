#!/bin/bash
# kernel-trace stats and HBM-side PMC passes of the bench in TZ_PREC_F16C8 (run on the GPU box from the repo root)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r02_f16c8 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-precision-report --precision f16c8 > $O/prof_r02_f16c8_bench.json 2> $O/prof_r02_f16c8.err || exit 1
export TZ_PRECISION=f16c8
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_r02_f16c8_fetch -- python3 tools/tower_only.py 6 > /dev/null 2>> $O/prof_r02_pmc.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_r02_f16c8_write -- python3 tools/tower_only.py 6 > /dev/null 2>> $O/prof_r02_pmc.err || exit 1
python3 tools/pmc_summary.py net_mfma_kernel $O/pmc_r02_f16c8_fetch $O/pmc_r02_f16c8_write > $O/r02_f16c8_hbm_pmc.csv
unset TZ_PRECISION
f=$(find $O/prof_r02_f16c8 -name "*kernel_stats.csv" | head -1)
cp "$f" $O/prof_r02_f16c8_kernel_stats.csv
head -6 $O/prof_r02_f16c8_kernel_stats.csv; cat $O/r02_f16c8_hbm_pmc.csv; cat $O/prof_r02_f16c8_bench.json | cut -c1-600
