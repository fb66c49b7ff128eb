cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_net.py -m gpu -q -x -k "f16x2" 2>&1 | tail -3
export TZ_PRECISION=f16x2
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_r02_f16x2_sq_b -- python3 tools/tower_only.py 6 > /dev/null 2>> gpurun_out/prof_r02_pmc.err
python3 tools/pmc_summary.py net_mfma_kernel gpurun_out/pmc_r02_f16x2_sq_b
unset TZ_PRECISION
python3 tools/precision_report.py 32 4096 40 2>&1 | tail -6
