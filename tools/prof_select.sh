#!/bin/bash
# SQ counters of the fused kernel (and of k_post_sel<double> in the two-kernel form) alone, for the library PSS_LIBRARY names:
#   PSS_LIBRARY=pyspecsdr_amd/libpss_nocompact.so bash tools/prof_select.sh gpurun_out/pmc_sel_old
set -u
OUT=${1:-gpurun_out/pmc_sel}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
for V in "" "fuse_post=0"; do
  T=f${V: -1}
  CMD="python tools/run_cells_alone.py 65536 $V"
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES \
      --kernel-trace --output-format csv -d "$OUT" -o a$T -- $CMD > "$OUT/a$T.log" 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS \
      --kernel-trace --output-format csv -d "$OUT" -o b$T -- $CMD > "$OUT/b$T.log" 2>&1
  python tools/pmc_summary.py "$OUT"/a${T}_counter_collection.csv "$OUT"/b${T}_counter_collection.csv | grep -A20 "k_spectrum_post\|k_post_sel" | grep -v "k_slide\|k_disp" 
done
