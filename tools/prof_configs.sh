#!/bin/bash
# HBM traffic of the other single-GPU configs (bench.py "other_configs"): one FETCH_SIZE and one WRITE_SIZE pass per config
# (separate --pmc passes, kernel trace only), each over `python tools/bench_configs.py <cfg> --launch-only` = two plain passes.
# tools/prof_configs_digest.py then adds {"configs": {cfg: {traffic_bytes, kernels}}} to the round's hbm_traffic.json.
#   bash tools/prof_configs.sh r04        (after tools/prof_round.sh r04: it extends gpurun_out/prof_r04/hbm_traffic.json)
set -u
TAG=${1:-r04}
OUT=gpurun_out/prof_cfgs_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
for CFG in cfg2_f32_rows cfg2_exact_cells cfg3 cfg4 cfg5_resident wfm_step; do
    CMD="python tools/bench_configs.py $CFG --launch-only"
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o ${CFG}_kt -- $CMD > "$OUT/${CFG}_kt.log" 2>&1
    timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT" -o ${CFG}_fetch -- $CMD > "$OUT/${CFG}_fetch.log" 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT" -o ${CFG}_write -- $CMD > "$OUT/${CFG}_write.log" 2>&1
done
python tools/prof_configs_digest.py "$OUT" "$TAG"
