#!/bin/bash
# Round 3 measurements of TZ_PREC_F16C6 in one call on the GPU box (from the repo root): kernel-trace stats of the bench in that
# precision, SQ_* / GRBM and FETCH_SIZE / WRITE_SIZE passes over tools/tower_only.py (separate --pmc passes, no other trace
# domain), the default bench line, config 4 (6x6) in both precisions with the 6x6 kernel's PMC pass.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03_f16c6 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-precision-report --precision f16c6 > $O/r03_f16c6_bench_line_profiled.json 2> $O/prof_r03_f16c6.err || exit 1
f=$(find $O/prof_r03_f16c6 -name "*kernel_stats.csv" | head -1)
cp "$f" $O/r03_f16c6_bench_kernel_stats.csv
head -5 $O/r03_f16c6_bench_kernel_stats.csv
export TZ_PRECISION=f16c6
bash tools/profile_c6.sh f16c6 net_c6_kernel > $O/r03_f16c6_net_pmc.txt || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_r03_f16c6_fetch -- python3 tools/tower_only.py 6 > /dev/null 2>> $O/prof_r03_pmc.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_r03_f16c6_write -- python3 tools/tower_only.py 6 > /dev/null 2>> $O/prof_r03_pmc.err || exit 1
python3 tools/pmc_summary.py net_c6_kernel $O/pmc_r03_f16c6_fetch $O/pmc_r03_f16c6_write > $O/r03_f16c6_hbm_pmc.csv || exit 1
cat $O/r03_f16c6_net_pmc.txt $O/r03_f16c6_hbm_pmc.csv
# config 4 (6x6, 2048 games, 800 simulations per move) in the tolerance precision and in the fp16 default
python3 tools/run_configs.py 4 > $O/r03_config4_f16c6_line.json 2> $O/r03_config4.err || exit 1
python3 tools/net6_bench.py 2048 100 > $O/r03_net6_f16c6_bench.json 2>&1 || exit 1
cat > $O/net6_only.py <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import takzero_amd.api as A
from takzero_amd import weights as W
net = A.Net(arch=A.ARCH_NET6_SIMHASH); net.load_tensors(W.init_weights(W.ARCH_NET6_SIMHASH, seed=123))
m = A.BatchedMCTS(2048, 6, 4, agent=net, node_capacity=2048); m.new_openings(np.arange(2048) % 16); m.simulate(np.zeros(2048, np.float32), 6); print(m.counters())
PY
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_r03_net6_c6 -- python3 $O/net6_only.py > /dev/null 2>> $O/prof_r03_pmc.err || exit 1
python3 tools/pmc_summary.py net_c6_kernel $O/pmc_r03_net6_c6 > $O/r03_net6_f16c6_pmc.csv
unset TZ_PRECISION
python3 tools/run_configs.py 4 > $O/r03_config4_f16_line.json 2>> $O/r03_config4.err || exit 1
cat $O/r03_config4_f16c6_line.json $O/r03_net6_f16c6_bench.json $O/r03_net6_f16c6_pmc.csv $O/r03_config4_f16_line.json
# the driver's own line
python3 bench.py > $O/r03_bench_line.json 2> $O/r03_bench.err || exit 1
cut -c1-1500 $O/r03_bench_line.json
