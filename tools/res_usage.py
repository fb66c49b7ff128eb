"""Table of the kernel-resource-usage remarks of one hipcc compile (hipcc ... -Rpass-analysis=kernel-resource-usage 2> file):
name, VGPRs, AGPRs, spilled VGPRs, scratch bytes per lane, waves per SIMD.  Usage: tools/res_usage.py file [filter]"""
import re
import subprocess
import sys

rows, cur = [], None
for line in open(sys.argv[1]):
    m = re.search(r"remark: ([A-Za-z \[\]/]+): (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
print("%-90s %5s %5s %6s %8s %4s" % ("kernel", "VGPR", "AGPR", "spill", "scratch", "occ"))
for r, nm in zip(rows, names):
    nm = nm.replace("(anonymous namespace)::", "").split("(")[0]
    if flt in nm:
        print("%-90s %5s %5s %6s %8s %4s" % (nm[:90], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"),
                                          r.get("Occupancy [waves/SIMD]")))
