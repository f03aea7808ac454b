#!/bin/bash
# bash tools/step_trace.sh TAG bench.py args...  ->  gpurun_out/step_TAG.csv (per-kernel time of ONE steady-state step)
# Steps are cut at the optimizer's launches: pass --no-fused-adam for LightGCN / NGCF (their default bench run applies the
# table's Adam update inside the last backward kernel, which leaves no launch to cut at).
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/steptrace_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 "$R/bench.py" "$@" > "$OUT.log" 2>&1 || exit 1
python3 "$R/tools/step_trace.py" "$OUT" "$R/gpurun_out/step_$TAG.csv" | tee "$R/gpurun_out/step_$TAG.txt"
rm -rf "$OUT"
