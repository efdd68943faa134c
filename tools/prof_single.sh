cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/single
for m in NFM AM WFM USB; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ts_$m -o t -- python $R/tools/trace_single.py $m 32768 10 > $R/gpurun_out/single/$m.log 2>&1
  f=$(find /tmp/ts_$m -name '*kernel_stats.csv' | head -1)
  cp "$f" $R/gpurun_out/single/${m}_kernel_stats.csv
done
