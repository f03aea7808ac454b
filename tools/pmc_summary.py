"""Fold the counter_collection.csv files of separate `rocprofv3 --pmc` passes into one per-kernel table.

    python tools/pmc_summary.py OUT_PREFIX DIR_FETCH DIR_WRITE DIR_TCC ["command line the passes ran"]

Writes OUT_PREFIX.csv and OUT_PREFIX.json (the file bench.py reads `roofline.traffic` from).  Only kernels whose name
contains `tagrec::` are kept.  gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE counts a
128-byte request as 64 bytes, so HBM-side bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024; the factor is re-checked on
`adam_kernel`, whose traffic is known exactly (16 B read + 12 B written per element)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"tagrec::([A-Za-z0-9_]+)(<[^>(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else None


def read(dirname):
    per = defaultdict(lambda: defaultdict(list))        # kernel -> counter -> [value per dispatch]
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                if k is None:
                    continue
                per[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    return per, dur


def main():
    out, d_fetch, d_write, d_tcc = sys.argv[1:5]
    cmd = sys.argv[5] if len(sys.argv) > 5 else ""
    fetch, dur = read(d_fetch)
    write, _ = read(d_write)
    tcc, _ = read(d_tcc)
    mean = lambda v: sum(v) / len(v) if v else 0.0
    rows, js = [], {}
    for k in fetch:
        f, w = mean(fetch[k].get("FETCH_SIZE", [])), mean(write.get(k, {}).get("WRITE_SIZE", []))
        hit, miss = mean(tcc.get(k, {}).get("TCC_HIT_sum", [])), mean(tcc.get(k, {}).get("TCC_MISS_sum", []))
        traffic = (2 * f + w) * 1024
        n = len(fetch[k]["FETCH_SIZE"])
        rows.append((k, n, f, w, hit, miss, traffic, mean(dur[k])))
        js[k] = {"launches": n, "fetch_kb_raw": f, "write_kb": w, "traffic_bytes_per_launch": traffic, "tcc_hit": hit, "tcc_miss": miss,
                 "mean_ms_under_pmc": mean(dur[k])}
    rows.sort(key=lambda r: -r[6] * r[1])
    with open(out + ".csv", "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(["kernel", "launches", "FETCH_SIZE_KB_raw", "WRITE_SIZE_KB", "TCC_HIT_sum", "TCC_MISS_sum",
                     "hbm_side_bytes_per_launch(2*FETCH+WRITE)*1024", "mean_ms_under_pmc"])
        for r in rows:
            wr.writerow([r[0], r[1], f"{r[2]:.1f}", f"{r[3]:.1f}", f"{r[4]:.0f}", f"{r[5]:.0f}", f"{r[6]:.0f}", f"{r[7]:.4f}"])
    import datetime
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("spmm.hip", "graph.h", "common.h"):
        with open(os.path.join(root, "tag-aware-recommendation_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    with open(out + ".json", "w") as fh:
        json.dump({"command": cmd, "collected": datetime.date.today().isoformat(), "head": os.environ.get("GRAFT_HEAD", "?"),
                   "kernel_source_sha16": h.hexdigest()[:16],
                   "source": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum TCC_MISS_sum> --kernel-trace, three separate passes of `"
                             + cmd + "` (1 MI355X)",
                   "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B -> doubled; WRITE_SIZE exact (check both on adam_kernel: "
                                 "16 B read and 12 B written per parameter element)",
                   "kernels": js}, fh, indent=1)
    print(f"wrote {out}.csv / .json ({len(rows)} kernels)")


if __name__ == "__main__":
    main()
