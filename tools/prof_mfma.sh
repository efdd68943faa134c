#!/bin/bash
# SQ counters of the exact and the matrix-pipe NFM forward kernels (GPU box, via gpurun)
set -u
OUT=${1:-gpurun_out/pmc_mfma}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
CMD="python tools/mfma_variant_stats.py 1"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES \
    --kernel-trace --output-format csv -d "$OUT" -o a -- $CMD > "$OUT/a.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    --kernel-trace --output-format csv -d "$OUT" -o b -- $CMD > "$OUT/b.log" 2>&1
python tools/pmc_summary.py "$OUT"/a_counter_collection.csv "$OUT"/b_counter_collection.csv | grep -A22 "k_nfm_fwd"
tail -2 "$OUT/a.log"
