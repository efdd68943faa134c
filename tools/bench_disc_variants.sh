#!/bin/bash
# The bench step under the three ways of feeding the NFM forward kernel its discriminator (default: inside the kernel; options
# "disc_rows": a pass of its own; "disc_spectrum": written by the 1024-point spectrum kernel) -> one JSON line each.
#   bash tools/bench_disc_variants.sh > gpurun_out/disc_variants.json
for o in "" "disc_rows=1" "disc_spectrum=1"; do
  PSS_OPTIONS=$o python bench.py --steps 50 --no-cpu-baseline --no-side 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'options': '$o' or 'default', 'ms_per_step': round(d['ms_per_step'], 4), 'value': round(d['value']), 'kernel_ms': d['roofline']['kernel_ms']}))"
done
