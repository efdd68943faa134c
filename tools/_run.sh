cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "post or pipeline or fused or cells or display or float64 or persistence or stream or spectrogram or waterfall or compaction" 2>&1 | tail -5
for L in "" nocompact ""; do
  if [ -n "$L" ]; then export PSS_LIBRARY=pyspecsdr_amd/libpss_$L.so; else unset PSS_LIBRARY; fi
  echo "== lib [$L]"; FUSE_VARIANTS=two64,fus32 timeout 600 python tools/ab_fuse.py 2>&1 | grep -E "sha256|vs|ms" | sed -n '1,2p;5,10p'
done
