#!/bin/bash
# SQ counters + HBM traffic of ONE config of tools/bench_configs.py on the GPU box (separate --pmc passes, kernel trace only).
#   bash tools/prof_cfg.sh cfg3 [outdir]
set -u
CFG=${1:-cfg3}
OUT=${2:-gpurun_out/pmc_$CFG}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
CMD="python tools/bench_configs.py $CFG --launch-only"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS \
    --kernel-trace --output-format csv -d "$OUT" -o a -- $CMD > "$OUT/a.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES \
    --kernel-trace --output-format csv -d "$OUT" -o b -- $CMD > "$OUT/b.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_CVT \
    --kernel-trace --output-format csv -d "$OUT" -o c -- $CMD > "$OUT/c.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT" -o fetch -- $CMD > "$OUT/fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT" -o write -- $CMD > "$OUT/write.log" 2>&1
python tools/pmc_summary.py "$OUT"/a_counter_collection.csv "$OUT"/b_counter_collection.csv "$OUT"/c_counter_collection.csv "$OUT"/fetch_counter_collection.csv "$OUT"/write_counter_collection.csv | tee "$OUT/summary.txt"
