cd $GRAFT_REPO_ROOT
HASH=$(python -c "import sys; sys.path.insert(0, 'tools'); import bench_configs as b; print(b.source_hash())")
{
echo "# extended campaign at source hash $HASH: tools/fuzz_gpu_vs_oracle.py FUZZ_SEED=41..48 x 300 cases, FUZZ_EDGE=1 FUZZ_SEED=51..53 x 200; tools/fuzz_gpu_vs_oracle2.py FUZZ_SEED=61..63 x 150: GPU float64 audio / int16 PCM / dB rows against the CPU oracle"
for s in 41 42 43 44 45 46 47 48; do echo -n "seed $s: "; FUZZ_SEED=$s timeout 900 python tools/fuzz_gpu_vs_oracle.py 300 2>&1 | tail -1; done
for s in 51 52 53; do echo -n "edge seed $s: "; FUZZ_EDGE=1 FUZZ_SEED=$s timeout 900 python tools/fuzz_gpu_vs_oracle.py 200 2>&1 | tail -1; done
for s in 61 62 63; do echo -n "fuzz2 seed $s: "; FUZZ_SEED=$s timeout 900 python tools/fuzz_gpu_vs_oracle2.py 150 2>&1 | tail -1; done
} > gpurun_out/fuzz_big.txt 2>&1
tail -16 gpurun_out/fuzz_big.txt
{
echo "# tools/fuzz_pipeline_f64.py FUZZ_SEED=71..76 x 150 cases: pss_frame_pipeline_f64 (random frame length, batch, history, display geometry, mode, signal, cut + halo) — cells, extremes, NFM PCM against the oracle's step from IQ"
for s in 71 72 73 74 75 76; do echo -n "pipeline seed $s: "; FUZZ_SEED=$s timeout 900 python tools/fuzz_pipeline_f64.py 150 2>&1 | tail -1; done
} > gpurun_out/fuzz_pipe.txt 2>&1
cat gpurun_out/fuzz_pipe.txt
