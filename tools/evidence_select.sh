#!/bin/bash
# The select's own evidence (profiles/<tag>_select_counters.txt, <tag>_fuzz_select.txt), on the GPU box through gpurun.  In the build container first:
#   python tools/build_variant.py nocompact -DPSS_POST_NO_COMPACT       (the previous search, same source)
#   gpurun -- 'bash tools/evidence_select.sh'    then:  cp gpurun_out/sel_pmc.txt profiles/r06_select_counters.txt; cp gpurun_out/fuzz_select.txt profiles/r06_fuzz_select.txt
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
HASH=$(python -c "import sys; sys.path.insert(0, 'tools'); import bench_configs as b; print(b.source_hash())")
{
echo "# tools/prof_select.sh at source hash $HASH: SQ counters of k_spectrum_post (display half alone) and of k_post_sel<double> (fuse_post=0), 65 536 x 1024 FM IQ, 4 launches each:"
echo "# the product library (compaction select, pss_post.h select_kth64_compact) and a -DPSS_POST_NO_COMPACT build of the same source; 'per wave' = per 16 rows"
echo "=== product"; bash tools/prof_select.sh gpurun_out/pmc_sel_new
echo "=== nocompact"; PSS_LIBRARY=pyspecsdr_amd/libpss_nocompact.so bash tools/prof_select.sh gpurun_out/pmc_sel_old
} > gpurun_out/sel_pmc.txt 2>&1
{
echo "# tools/fuzz_select.py at source hash $HASH: the post-process of random dB rows (ten families: noise, grids, runs, two modes, clusters inside one high word, skewed, NaN / inf, constant, narrow, grouped), row lengths 8 ... 4096, float64 rows and float32 rows, against np.convolve's sums / np.median / clamp / finite extremes bit for bit"
for s in 1 2 3 4 5 6 7 8; do echo -n "seed $s: "; FUZZ_SEED=$s timeout 900 python tools/fuzz_select.py 150 2>&1 | grep -v amdgpu.ids | tail -3; done
echo "# the same rows through a -DPSS_POST_NO_COMPACT build (the previous search)"
for s in 1 2; do echo -n "seed $s (nocompact): "; PSS_LIBRARY=pyspecsdr_amd/libpss_nocompact.so FUZZ_SEED=$s timeout 900 python tools/fuzz_select.py 150 2>&1 | grep -v amdgpu.ids | tail -3; done
} > gpurun_out/fuzz_select.txt 2>&1
tail -3 gpurun_out/fuzz_select.txt
