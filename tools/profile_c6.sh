cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export TZ_PRECISION=${1:-f16c6}
PAT=${2:-net_c6_kernel}
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_${TZ_PRECISION}_a -- python3 tools/tower_only.py 6 > /dev/null 2>> gpurun_out/prof_c6_pmc.err
python3 tools/pmc_summary.py $PAT gpurun_out/pmc_${TZ_PRECISION}_a
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_${TZ_PRECISION}_b -- python3 tools/tower_only.py 6 > /dev/null 2>> gpurun_out/prof_c6_pmc.err
python3 tools/pmc_summary.py $PAT gpurun_out/pmc_${TZ_PRECISION}_b
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc_${TZ_PRECISION}_c -- python3 tools/tower_only.py 6 > /dev/null 2>> gpurun_out/prof_c6_pmc.err
python3 tools/pmc_summary.py $PAT gpurun_out/pmc_${TZ_PRECISION}_c
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/pmc_${TZ_PRECISION}_d -- python3 tools/tower_only.py 6 > /dev/null 2>> gpurun_out/prof_c6_pmc.err
python3 tools/pmc_summary.py $PAT gpurun_out/pmc_${TZ_PRECISION}_d
