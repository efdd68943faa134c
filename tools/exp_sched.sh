# schedule experiments of pss_frame_pipeline_nfm (variant build with -DPSS_VARIANTS: option pipe_sched)
for o in "" "pipe_sched=3" "pipe_sched=5" "pipe_sched=4"; do
  echo "== $o"; PSS_LIBRARY=pyspecsdr_amd/libpss_variants.so PSS_OPTIONS="$o" timeout 300 python bench.py --steps 30 --no-side --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['verified']['ok'], d['roofline']['kernel_ms'])"
done
