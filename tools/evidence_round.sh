#!/bin/bash
# Everything profiles/<tag>_* holds, in one gpurun call, all from ONE source tree (every file is stamped with its hash):
#   bash tools/evidence_round.sh r04 [quick]     then, back in the build container:  bash tools/copy_profiles.sh r04
# 1. tools/prof_round.sh   : kernel trace + HBM / SQ counters of the default bench command, standalone launches -> digest
# 2. tools/prof_configs.sh : per-config traffic of cfg 3 / 4 / 5 / WFM                                          -> digest["configs"]
# 3. the default bench line WITH the digest of (1, 2) in place (so roofline.traffic / other_configs.*.traffic_ratio are filled)
# 4. shim latency table, GPU-vs-oracle fuzzers (skipped with "quick")
set -u
TAG=${1:-r04}
QUICK=${2:-}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python tools/kernel_resources.py > /dev/null      # the code object's registers / spills at THIS source, before any summary cites it
bash tools/prof_round.sh "$TAG" > gpurun_out/prof_round.log 2>&1
bash tools/prof_configs.sh "$TAG" > gpurun_out/prof_configs.log 2>&1
cp "gpurun_out/prof_$TAG/hbm_traffic.json" profiles/hbm_traffic.json
timeout 900 python bench.py 2> gpurun_out/bench.err > gpurun_out/bench_line.json
HASH=$(python -c "import sys; sys.path.insert(0, 'tools'); import bench_configs as b; print(b.source_hash())")
if [ -z "$QUICK" ]; then
    timeout 600 python tools/shim_latency.py > gpurun_out/shim_latency.txt 2>&1
    # where a single read buffer's time goes: rocprofv3 kernel stats of ten calls per mode, and the lone-wavefront issue facts behind them
    bash tools/prof_single.sh > /dev/null 2>&1
    {
        echo "# one 32 768-sample read buffer through demodulate_signal, source hash $HASH: rocprofv3 --kernel-trace --stats of tools/trace_single.py (11 calls per mode)"
        python tools/single_digest.py gpurun_out/single
        echo "# tools/ubench/exec_mask.hip: does a float64 instruction cost less with fewer active lanes?  (no: a lone wavefront pays ~4.4-5.4 clk per float64"
        echo "# instruction and 9-10 clk per dependent one whatever the EXEC mask; the 11-instruction biquad step of the systolic kernels = 48 clk per sample)"
        hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/exec_mask.hip -o /tmp/exec_mask 2>/dev/null && /tmp/exec_mask
    } > gpurun_out/single_buffer.txt 2>&1
    {
        echo "# tools/fuzz_gpu_vs_oracle.py on the MI355X box at source hash $HASH: FUZZ_SEED=11,12,13 x 120 cases and FUZZ_EDGE=1 FUZZ_SEED=21 x 100 cases; tools/fuzz_gpu_vs_oracle2.py x 60: float64 audio and int16 PCM against the CPU oracle, bit for bit"
        for s in 11 12 13; do FUZZ_SEED=$s timeout 600 python tools/fuzz_gpu_vs_oracle.py 120 2>&1 | tail -1; done
        FUZZ_EDGE=1 FUZZ_SEED=21 timeout 600 python tools/fuzz_gpu_vs_oracle.py 100 2>&1 | tail -1
        FUZZ_SEED=31 timeout 600 python tools/fuzz_gpu_vs_oracle2.py 60 2>&1 | tail -1
    } > gpurun_out/fuzz_gpu.txt
fi
tail -3 "gpurun_out/prof_cfgs_$TAG/${TAG}_configs_summary.txt"
python -c "
import json; d = json.load(open('gpurun_out/bench_line.json'))
print(d['value'], d['ms_per_step'], d['roofline'].get('traffic'), {k: (v.get('ms'), v.get('traffic_ratio')) for k, v in d['other_configs'].items()})"
