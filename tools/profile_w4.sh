cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export TZ_PRECISION=f16
for w in 0 1; do
export TZ_NET_W4=$w
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_w4_${w}_a -- python3 tools/tower_only.py 6 > /dev/null 2>> gpurun_out/prof_w4_pmc.err
python3 tools/pmc_summary.py net_mfma_kernel gpurun_out/pmc_w4_${w}_a
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_w4_${w}_b -- python3 tools/tower_only.py 6 > /dev/null 2>> gpurun_out/prof_w4_pmc.err
python3 tools/pmc_summary.py net_mfma_kernel gpurun_out/pmc_w4_${w}_b
python3 - <<PY
import csv, glob
for f in glob.glob("gpurun_out/pmc_w4_${w}_a/**/*kernel_trace.csv", recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if "net_mfma" in r["Kernel_Name"]]
    d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows][1:]
    print("W4=${w} kernel ns avg", sum(d)/len(d), len(d))
PY
done
