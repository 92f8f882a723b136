"""Summarise a rocprofv3 `--kernel-trace --pmc ...` run (csv output): one line per dispatch of kernels whose
name contains the given substring, with every collected counter summed over the chip and the duration.
    python tools/pmc_summary.py <dir with *_counter_collection.csv + *_kernel_trace.csv> [name substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "gemm")
    cc = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)
    dur = {}
    for f in kt:
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    rows = defaultdict(dict)
    for f in cc:
        for r in csv.DictReader(open(f)):
            if sub not in r["Kernel_Name"]:
                continue
            e = rows[int(r["Dispatch_Id"])]
            e["kernel"] = r["Kernel_Name"][:48]
            e["grid"] = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for did in sorted(rows):
        e = rows[did]
        e["dur_us"] = dur.get(str(did))
        print("dispatch %d: %s" % (did, e))


if __name__ == "__main__":
    main()
