for o in "" "pipe_sched=4" "pipe_sched=1"; do
  echo "== $o"; PSS_LIBRARY=pyspecsdr_amd/libpss_variants.so PSS_OPTIONS="$o" timeout 300 python bench.py --steps 30 --no-side --no-cpu-baseline --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
done
