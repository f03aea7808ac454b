#!/bin/bash
# bash tools/pmc_microbench.sh TAG "COUNTERS A" "COUNTERS B" ... -- script.py args   (one rocprofv3 --pmc pass per counter group)
set -o pipefail
TAG=$1; shift
GROUPS_=()
while [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcmb_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for C in "${GROUPS_[@]}"; do
  D=$OUT/p$i
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$D" -- python3 "$R/$1" "${@:2}" > "$D.log" 2>&1 || { tail -5 "$D.log"; exit 1; }
  python3 - "$D" <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "tagrec" not in k: continue
    print(k, {c: f"{sum(v)/len(v):.4g}" for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
  rm -rf "$D"
  i=$((i+1))
done
