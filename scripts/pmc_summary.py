"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per launch, grouped by kernel name.

usage: python scripts/pmc_summary.py <dir-or-csv> [<dir-or-csv> ...] [--match SUBSTR] [--out file.csv]
Every *counter_collection.csv under the given paths is read (one --pmc pass each); kernels whose name contains
SUBSTR (default: all) are kept; template arguments are kept in the name so kernel variants stay apart."""
import argparse
import collections
import csv
import glob
import os
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("paths", nargs="+")
    ap.add_argument("--match", default="")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    files = []
    for p in a.paths:
        if os.path.isdir(p):
            files += sorted(glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True))
        else:
            files.append(p)
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, set()]))
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                if a.match and a.match not in k:
                    continue
                k = k.replace("void ", "").replace("(anonymous namespace)::", "")
                k = k.split("(")[0]
                cell = acc[k][row["Counter_Name"]]
                cell[0] += float(row["Counter_Value"])
                cell[1].add(row["Dispatch_Id"])
    counters = sorted({c for k in acc for c in acc[k]})
    out = open(a.out, "w") if a.out else sys.stdout
    out.write("kernel,launches," + ",".join(counters) + "\n")
    for k in sorted(acc):
        n = max(len(acc[k][c][1]) for c in acc[k])
        vals = ["%.6g" % (acc[k][c][0] / max(1, len(acc[k][c][1]))) if c in acc[k] else "" for c in counters]
        out.write('"%s",%d,%s\n' % (k, n, ",".join(vals)))


if __name__ == "__main__":
    main()
