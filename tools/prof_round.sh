#!/bin/bash
# Round profile on the GPU box (run through gpurun): everything bench.py quotes comes out of this one script.
#   1. rocprofv3 --kernel-trace --stats of the default bench command                  -> <tag>_kernel_stats.csv
#   2. HBM traffic counters, FETCH_SIZE and WRITE_SIZE in separate --pmc passes       -> digest
#   3. SQ counters (VALU activity, waits) and the float64 instruction mix              -> <tag>_sq_counters.txt
#   4. digest: profiles/<tag>_summary.txt + profiles/hbm_traffic.json, stamped with the git hash, the hash of the kernel
#      sources (bench.py only quotes the digest when that hash matches the tree it runs from) and each kernel's VGPR / LDS.
#   bash tools/prof_round.sh r02
set -u
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
CMD="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side --no-other-configs"
SHORT="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side --no-other-configs"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o kt -- $CMD > "$OUT/kt.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT" -o fetch -- $CMD > "$OUT/fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT" -o write -- $CMD > "$OUT/write.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVES \
    --kernel-trace --output-format csv -d "$OUT" -o sqa -- $SHORT > "$OUT/sqa.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --kernel-trace --output-format csv -d "$OUT" -o sqb -- $SHORT > "$OUT/sqb.log" 2>&1
# standalone launches of the step's kernels (inside a bench step the display chain and the demodulator share the machine)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o alone -- python tools/bench_alone.py > "$OUT/alone.log" 2>&1
python tools/prof_digest.py "$OUT" "$TAG"
