#!/usr/bin/env python3
"""Per-step kernel breakdown from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py.

    python3 tools/step_trace.py <dir with *_kernel_trace.csv> [out.csv]

A training step ends with a burst of `adam_kernel` / `adam_multi_kernel` launches, so the trace is cut at the end
of every burst; the LAST complete window that is preceded by another burst is one steady-state step.  Prints, for that
step, the time per kernel name (sum, calls, mean, max), the busy time and the idle gaps between kernels."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    path = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # bursts of adam kernels: consecutive adam launches with < 2 ms between them
    ends = []
    last = None
    for s, e, n in rows:
        if "adam_kernel" in n or "adam_multi_kernel" in n:
            if last is not None and s - last > 2_000_000:
                ends.append(last)
            last = e
    if last is not None:
        ends.append(last)
    which = int(os.environ.get("STEP", "-1"))
    if len(ends) < 2:
        sys.exit("fewer than two optimizer bursts in the trace (LightGCN / NGCF: run bench.py with --no-fused-adam, see step_trace.sh)")
    for i in range(1, len(ends)):
        n_in = sum(1 for s, e, n in rows if s > ends[i - 1] and e <= ends[i])
        print(f"# window {i - len(ends)}: {(ends[i] - ends[i - 1]) / 1e6:.2f} ms, {n_in} launches")
    hi = ends[which]
    lo = ends[which - 1]
    step = [(s, e, n) for s, e, n in rows if s > lo and e <= hi]
    if os.environ.get("WINDOW"):                       # WINDOW=a,b (ms from the start of the step): only that part of it
        a, b = (float(x) * 1e6 for x in os.environ["WINDOW"].split(","))
        step = [(s, e, n) for s, e, n in step if s - lo >= a and s - lo < b]
        lo, hi = lo + int(a), lo + int(b)
    agg = defaultdict(lambda: [0, 0, 0])
    busy, gap, prev = 0, 0, lo
    for s, e, n in step:
        a = agg[n]
        a[0] += e - s
        a[1] += 1
        a[2] = max(a[2], e - s)
        busy += e - s
        if s > prev:
            gap += s - prev
        prev = max(prev, e)
    wall = hi - lo
    # the largest idle gaps and the kernels on either side (GAPS=n lists n of them)
    n_gaps = int(os.environ.get("GAPS", "0"))
    if n_gaps:
        gaps, prev_e, prev_n = [], lo, "(previous step)"
        for s, e, n in step:
            if s > prev_e:
                gaps.append((s - prev_e, (prev_e - lo) / 1e6, prev_n[:70], n[:70]))
            if e > prev_e:
                prev_e, prev_n = e, n
        hist = [sum(g[0] for g in gaps if lo_ <= g[0] < hi_) / 1e6 for lo_, hi_ in ((0, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e12))]
        print(f"# idle by gap length: <5 us {hist[0]:.2f} ms, 5-20 us {hist[1]:.2f} ms, 20-100 us {hist[2]:.2f} ms, >100 us {hist[3]:.2f} ms")
        for g in sorted(gaps, reverse=True)[:n_gaps]:
            print(f"# gap {g[0] / 1e3:8.1f} us at +{g[1]:7.2f} ms  after {g[2]}  before {g[3]}")
    # TIMELINE=us: every kernel of the step at least that long, in launch order, with the number of shorter launches between
    tl = float(os.environ.get("TIMELINE", "0"))
    if tl:
        small_n, small_t = 0, 0
        for s, e, n in step:
            if e - s >= tl * 1e3:
                print(f"# +{(s - lo) / 1e6:7.2f} ms  {(e - s) / 1e3:8.1f} us  {n[:90]}   (after {small_n} shorter launches, {small_t / 1e3:.0f} us)")
                small_n, small_t = 0, 0
            else:
                small_n += 1
                small_t += e - s
    out = [("kernel", "total_ms", "calls", "mean_us", "max_us", "share")]
    for n, (t, c, m) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        out.append((n[:110], f"{t / 1e6:.3f}", c, f"{t / c / 1e3:.1f}", f"{m / 1e3:.1f}", f"{t / wall:.3f}"))
    print(f"# {path}\n# step window {wall / 1e6:.2f} ms, kernels busy {busy / 1e6:.2f} ms, idle gaps {gap / 1e6:.2f} ms, "
          f"{len(step)} launches, {len(ends)} optimizer bursts in the trace")
    w = csv.writer(open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout)
    w.writerows(out)


if __name__ == "__main__":
    main()
