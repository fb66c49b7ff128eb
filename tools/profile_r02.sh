#!/bin/bash
# Round-2 profiles of the default bench (run on the GPU box from the repo root; outputs under gpurun_out/prof_r02_*):
# kernel-trace stats of bench.py (fp16 default and f16x2), then the PMC passes (FETCH_SIZE, WRITE_SIZE, SQ_* + GRBM in separate
# runs, never combined with a trace domain other than --kernel-trace) over tools/tower_only.py.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r02_f16 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-precision-report > $O/prof_r02_f16_bench.json 2> $O/prof_r02_f16.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r02_f16x2 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-precision-report --precision f16x2 > $O/prof_r02_f16x2_bench.json 2> $O/prof_r02_f16x2.err || exit 1
for prec in f16 f16x2; do
  export TZ_PRECISION=$prec
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_r02_${prec}_fetch -- python3 tools/tower_only.py 6 > /dev/null 2>> $O/prof_r02_pmc.err || exit 1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_r02_${prec}_write -- python3 tools/tower_only.py 6 > /dev/null 2>> $O/prof_r02_pmc.err || exit 1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_r02_${prec}_sq -- python3 tools/tower_only.py 6 > /dev/null 2>> $O/prof_r02_pmc.err || exit 1
  python3 tools/pmc_summary.py net_mfma_kernel $O/pmc_r02_${prec}_fetch $O/pmc_r02_${prec}_write $O/pmc_r02_${prec}_sq > $O/r02_${prec}_net_pmc.csv
done
unset TZ_PRECISION
for d in $O/prof_r02_f16 $O/prof_r02_f16x2; do
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/$(basename $d)_kernel_stats.csv
done
head -5 $O/prof_r02_f16_kernel_stats.csv; head -5 $O/prof_r02_f16x2_kernel_stats.csv; cat $O/r02_f16_net_pmc.csv $O/r02_f16x2_net_pmc.csv
