#!/bin/bash
# Collect two SQ counter passes over a short bench run on the GPU box and print per-kernel summaries.
# usage (on the GPU box, via gpurun):  bash tools/prof_pmc.sh [outdir]
set -u
OUT=${1:-gpurun_out/pmc}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS \
    --kernel-trace --output-format csv -d "$OUT" -o a -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/a.log" 2>&1
timeout 500 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES \
    --kernel-trace --output-format csv -d "$OUT" -o b -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/b.log" 2>&1
python tools/pmc_summary.py "$OUT/a_counter_collection.csv" "$OUT/b_counter_collection.csv"
