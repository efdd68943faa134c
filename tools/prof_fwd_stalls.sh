#!/bin/bash
# Where do the wavefronts of k_nfm_fwd wait?  Three counter passes over a plain NFM demodulate call (tools/time_fwd.py); prints the
# mean of every counter over the kernel's large launches.   bash tools/prof_fwd_stalls.sh
set -u
OUT=gpurun_out/prof_fwd_stalls
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
P1="SQ_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P2="SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD"
P3="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR"
P4="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_BRANCH"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i + 1))
    timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT" -o p$i -- python tools/time_fwd.py 1 > "$OUT/p$i.log" 2>&1
done
python - "$OUT" <<'PY'
import csv, collections, sys, os
out = sys.argv[1]
acc = collections.defaultdict(list)
for i in (1, 2, 3, 4):
    p = os.path.join(out, f"p{i}_counter_collection.csv")
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        if "k_nfm_fwd" in r["Kernel_Name"] and float(r["Grid_Size"]) >= 262144:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(f"{k:30s} {sum(acc[k]) / len(acc[k]):16.5e}  ({len(acc[k])} launches)")
PY
