#!/bin/bash
# Round profile on the GPU box (run through gpurun): kernel-trace stats of the default bench command, plus HBM
# traffic counters in separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
#   bash tools/prof_round.sh r01
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
CMD="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o kt -- $CMD > "$OUT/kt.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT" -o fetch -- $CMD > "$OUT/fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT" -o write -- $CMD > "$OUT/write.log" 2>&1
python tools/prof_digest.py "$OUT" "$TAG"
