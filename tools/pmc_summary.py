"""Summarises rocprofv3 --pmc CSV output (one directory per pass, as written by
tools/pmc_passes.sh) for one kernel: sums each counter over the dispatches of
that kernel and prints per-dispatch averages as JSON.

    python tools/pmc_summary.py gpurun_out/pmc render_fast_kernel > profiles/pmc_r01.json
"""
import csv
import glob
import json
import os
import sys


def main():
    root, kernel = sys.argv[1], sys.argv[2]
    out = {}
    for path in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
        pass_name = os.path.relpath(path, root).split(os.sep)[0]
        sums, disp = {}, set()
        with open(path) as f:
            for row in csv.DictReader(f):
                if kernel not in row["Kernel_Name"]:
                    continue
                disp.add(row["Dispatch_Id"])
                sums[row["Counter_Name"]] = sums.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        n = max(1, len(disp))
        out[pass_name] = {"dispatches": len(disp), "per_dispatch": {k: v / n for k, v in sorted(sums.items())}}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
