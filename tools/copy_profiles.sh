#!/bin/bash
# Copy the digests of the last tools/prof_round.sh / prof_other.sh / shim_latency.py / bench_configs.py cfg5stream runs
# (merged back under gpurun_out/) into profiles/, the tracked place the bench line and the docs cite.
#   bash tools/copy_profiles.sh r02
set -eu
TAG=${1:-r02}
cd "$(dirname "$0")/.."
cp gpurun_out/prof_$TAG/${TAG}_summary.txt gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv gpurun_out/prof_$TAG/${TAG}_sq_counters.txt profiles/
cp gpurun_out/prof_$TAG/hbm_traffic.json profiles/hbm_traffic.json
[ -f gpurun_out/prof_other_$TAG/${TAG}_summary.txt ] && cp gpurun_out/prof_other_$TAG/${TAG}_summary.txt profiles/${TAG}_other_configs_kernel_stats_and_hbm.txt
# tools/prof_configs.sh: every config of bench.py's other_configs, per-config counters (supersedes prof_other.sh's file)
[ -f gpurun_out/prof_cfgs_$TAG/${TAG}_configs_summary.txt ] && cp gpurun_out/prof_cfgs_$TAG/${TAG}_configs_summary.txt profiles/${TAG}_other_configs_kernel_stats_and_hbm.txt
[ -f gpurun_out/fuzz_gpu.txt ] && grep -v "amdgpu.ids" gpurun_out/fuzz_gpu.txt > profiles/${TAG}_fuzz_gpu_vs_oracle.txt
[ -f gpurun_out/shim_latency.txt ] && grep -v "amdgpu.ids" gpurun_out/shim_latency.txt > profiles/${TAG}_shim_latency.txt
[ -f gpurun_out/single_buffer.txt ] && grep -v "amdgpu.ids\|warning\|nodiscard" gpurun_out/single_buffer.txt > profiles/${TAG}_single_buffer.txt
[ -f gpurun_out/prof_xl_$TAG/${TAG}_xl_sq_counters.txt ] && cp gpurun_out/prof_xl_$TAG/${TAG}_xl_sq_counters.txt profiles/
[ -f gpurun_out/cfg5stream.txt ] && grep '^{"config"' gpurun_out/cfg5stream.txt > profiles/${TAG}_cfg5_streamed.json
[ -f gpurun_out/bench_line.json ] && cp gpurun_out/bench_line.json profiles/${TAG}_bench_line.json
git status --short profiles
