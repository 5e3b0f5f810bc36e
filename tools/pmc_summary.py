"""Aggregate rocprofv3 --pmc counter_collection CSVs: per kernel and counter, the mean over launches of the
per-dispatch total (rows of one dispatch -- one per XCD / instance -- are summed).
usage: pmc_summary.py out.json dir_or_csv [dir_or_csv ...]"""
import csv, glob, json, os, sys
from collections import defaultdict

def main():
    out, srcs = sys.argv[1], sys.argv[2:]
    files = []
    for s in srcs:
        files += [s] if s.endswith(".csv") else glob.glob(os.path.join(s, "**", "*counter_collection.csv"), recursive=True)
    per = defaultdict(lambda: defaultdict(float))      # (file, kernel, dispatch) -> counter -> total
    for f in files:
        for row in csv.DictReader(open(f)):
            k = (f, row["Kernel_Name"].split("(")[0], row["Dispatch_Id"])
            per[k][row["Counter_Name"]] += float(row["Counter_Value"])
    agg = defaultdict(lambda: defaultdict(list))
    for (f, kern, _), ctrs in per.items():
        for c, v in ctrs.items():
            agg[kern][c].append(v)
    res = {kern: {c: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in sorted(ctrs.items())}
           for kern, ctrs in sorted(agg.items())}
    json.dump(res, open(out, "w"), indent=1)
    for kern, ctrs in res.items():
        if "mfma" in kern:
            print(kern, {c: round(d["mean_per_launch"], 1) for c, d in ctrs.items()})

if __name__ == "__main__":
    main()
