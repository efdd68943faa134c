#!/bin/bash
# SQ counters of the spectrum kernels that sit furthest from the HBM roof (k_spectrum_xl: scanner slices 8192 x 4096, compute_fft rows
# 16384 x 8192 and 8192 x 16384), one shape per process so that every dispatch of a csv belongs to it:
#   bash tools/prof_xl.sh r05   ->  gpurun_out/prof_xl_r05/r05_xl_sq_counters.txt
set -u
TAG=${1:-r05}
OUT=gpurun_out/prof_xl_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
SUM="$OUT/${TAG}_xl_sq_counters.txt"
echo "== SQ counters per launch of k_spectrum_xl (mean over dispatches; SQ_* cycle counters are in units of 4 clocks, summed over the waves)" > "$SUM"
for SHAPE in "4096 8192 scan" "8192 16384" "16384 8192" "4096 32768" "1024 65536"; do
    NAME=$(echo $SHAPE | tr ' ' '_')
    echo "== python tools/run_spec.py $SHAPE" >> "$SUM"
    python tools/run_spec.py $SHAPE 2>/dev/null | tail -1 >> "$SUM"
    timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES \
        --kernel-trace --output-format csv -d "$OUT" -o ${NAME}_a -- python tools/run_spec.py $SHAPE 3 > "$OUT/${NAME}_a.log" 2>&1
    timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
        --kernel-trace --output-format csv -d "$OUT" -o ${NAME}_b -- python tools/run_spec.py $SHAPE 3 > "$OUT/${NAME}_b.log" 2>&1
    timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES \
        --kernel-trace --output-format csv -d "$OUT" -o ${NAME}_c -- python tools/run_spec.py $SHAPE 3 > "$OUT/${NAME}_c.log" 2>&1
    python tools/pmc_summary.py "$OUT"/${NAME}_[abc]_counter_collection.csv >> "$SUM" 2>&1
done
cat "$SUM"
