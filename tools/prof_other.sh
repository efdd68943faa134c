#!/bin/bash
# Kernel-trace stats + HBM counters of the widened rows (WFM dispatcher path, cfg 3 AM/SSB/power/16k spectrum) on the GPU box.
#   bash tools/prof_other.sh r01
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_other_${TAG}${3:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
CMD=${2:-"python tools/bench_configs.py wfm cfg3"}   # bash tools/prof_other.sh r01h "python tools/bench_configs.py classify sweep"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o kt -- $CMD > "$OUT/kt.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT" -o fetch -- $CMD > "$OUT/fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT" -o write -- $CMD > "$OUT/write.log" 2>&1
python tools/prof_digest.py "$OUT" "$TAG" > /dev/null
sed -i "s#python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side#$CMD#" "$OUT/${TAG}_summary.txt"
grep '^{"config"' "$OUT/kt.log" >> "$OUT/${TAG}_summary.txt"
cat "$OUT/${TAG}_summary.txt"
