cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/net6_bench.py 2048 100 > gpurun_out/r2_net6_bench.json 2>&1; cat gpurun_out/r2_net6_bench.json
cat > gpurun_out/net6_only.py <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import takzero_amd.api as A
from takzero_amd import weights as W
net = A.Net(arch=A.ARCH_NET6_SIMHASH); net.load_tensors(W.init_weights(W.ARCH_NET6_SIMHASH, seed=123))
m = A.BatchedMCTS(2048, 6, 4, agent=net, node_capacity=2048); m.new_openings(np.arange(2048) % 16); m.simulate(np.zeros(2048, np.float32), 6); print(m.counters())
PY
for f in 8 4; do
  if [ $f = 4 ]; then export TZ_NET_P6=4; else unset TZ_NET_P6; fi
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_r02_net6_p$f -- python3 gpurun_out/net6_only.py > /dev/null 2>> gpurun_out/pmc6.err
  python3 tools/pmc_summary.py net_mfma_kernel gpurun_out/pmc_r02_net6_p$f
done
