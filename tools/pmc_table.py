"""Per-kernel means of whatever counters one or more `rocprofv3 --pmc ... --kernel-trace --output-format csv` passes collected.
    python tools/pmc_table.py DIR [DIR ...]      (tagrec:: kernels only)"""
import csv, glob, os, re, sys
from collections import defaultdict

vals = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                m = re.search(r"tagrec::([A-Za-z0-9_]+)(<[^>(]*>)?", row["Kernel_Name"])
                if not m:
                    continue
                k = m.group(1) + (m.group(2) or "")
                vals[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                vals[k]["_ms"].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
names = sorted({c for k in vals for c in vals[k]})
print("kernel," + ",".join(names))
for k, cs in sorted(vals.items(), key=lambda kv: -sum(kv[1]["_ms"])):
    print(k + "," + ",".join(f"{sum(cs[c]) / len(cs[c]):.4g}" if cs.get(c) else "" for c in names))
