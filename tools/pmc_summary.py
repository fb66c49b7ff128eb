"""Mean per launch of every counter in rocprofv3 --pmc output directories, for kernels whose name contains a pattern
(the first launch of each kernel is dropped).  `python tools/pmc_summary.py <pattern> <dir> [<dir> ...]`"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    pat = sys.argv[1]
    for d in sys.argv[2:]:
        vals = defaultdict(lambda: defaultdict(float))   # counter -> dispatch id -> value (summed over XCDs / SEs)
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    if pat in row["Kernel_Name"]:
                        vals[row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
        for counter, per in sorted(vals.items()):
            ids = sorted(per)[1:]
            if ids:
                print("%s,%s,%.1f,%d launches" % (os.path.basename(d.rstrip("/")), counter, sum(per[i] for i in ids) / len(ids), len(ids)))


if __name__ == "__main__":
    main()
