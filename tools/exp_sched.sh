# schedule experiments of pss_frame_pipeline_nfm (variant build with -DPSS_VARIANTS: option pipe_sched; the product's pipe_overlap)
#   python tools/build_variant.py variants -DPSS_VARIANTS && gpurun -- 'bash tools/exp_sched.sh'
for o in "" "pipe_sched=1" "pipe_sched=6" "pipe_sched=4" "pipe_sched=3" "pipe_sched=5" "pipe_overlap=2" "pipe_overlap=3" "" "pipe_sched=1"; do
  echo "== $o"; PSS_LIBRARY=pyspecsdr_amd/libpss_variants.so PSS_OPTIONS="$o" timeout 300 python bench.py --steps 30 --no-side --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['verified']['ok'], d['roofline']['kernel_ms'])"
done
