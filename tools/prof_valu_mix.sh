#!/bin/bash
# Dynamic VALU instruction mix of the bench kernels (float64 add / mul / fma, float32, conversions, integer) on the GPU box.
set -u
OUT=${1:-gpurun_out/valumix}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_WAVES \
    --kernel-trace --output-format csv -d "$OUT" -o a -- $CMD > "$OUT/a.log" 2>&1
timeout 500 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_LDS \
    --kernel-trace --output-format csv -d "$OUT" -o b -- $CMD > "$OUT/b.log" 2>&1
python tools/pmc_summary.py "$OUT/a_counter_collection.csv" "$OUT/b_counter_collection.csv"
