#!/bin/bash
# SQ counters of the post-process kernel alone (GPU box, via gpurun):  bash tools/prof_post.sh [outdir]
set -u
OUT=${1:-gpurun_out/pmc_post}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
CMD="python tools/bench_post.py 65536 1024"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES \
    --kernel-trace --output-format csv -d "$OUT" -o a -- $CMD > "$OUT/a.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    --kernel-trace --output-format csv -d "$OUT" -o b -- $CMD > "$OUT/b.log" 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS \
    --kernel-trace --output-format csv -d "$OUT" -o c -- $CMD > "$OUT/c.log" 2>&1
python tools/pmc_summary.py "$OUT"/a_counter_collection.csv "$OUT"/b_counter_collection.csv "$OUT"/c_counter_collection.csv
python - <<PY
import csv
for r in csv.DictReader(open("$OUT/a_kernel_trace.csv")):
    if "k_post" in r["Kernel_Name"]:
        print(r["Kernel_Name"][:60], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, "us")
PY
