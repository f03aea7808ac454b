#!/bin/bash
# Refresh the rocprofv3 summaries of one bench.py command on the GPU box (run through gpurun from the repo root):
#     bash tools/profile.sh TAG bench.py args...        e.g.  bash tools/profile.sh c2_lightgcn --steps 3 --warmup 1 --no-cpu --big-batch 0
# Leaves gpurun_out/prof_TAG/{stats.csv,pmc.csv,pmc.json}; copy what should be judged into profiles/.
# One --kernel-trace --stats pass, then the three counter passes on their own (never combined with other tracing).
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/bench.py" "$@" > "$OUT/stats.log" 2>&1 || exit 1
cp "$(ls "$OUT"/stats/*/*kernel_stats.csv | head -1)" "$OUT/stats.csv"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  D=$OUT/pmc_$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$D" -- python3 "$R/bench.py" "$@" > "$D.log" 2>&1 || exit 1
done
python3 "$R/tools/pmc_summary.py" "$OUT/pmc" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "$OUT/pmc_TCC_HIT_sum_TCC_MISS_sum" "python3 bench.py $*"
# keep only the summaries (the raw traces are large)
rm -rf "$OUT"/stats "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE "$OUT"/pmc_TCC_HIT_sum_TCC_MISS_sum
head -12 "$OUT/stats.csv" | cut -c1-160
