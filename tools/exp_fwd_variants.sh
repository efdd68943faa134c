# timing experiments on k_nfm_fwd: variant builds (tools/build_variant.py <name> -DPSS_EXP_...) against the product library, same box
for v in "" $@; do
  if [ -z "$v" ]; then python tools/time_fwd.py 2; else PSS_LIBRARY=pyspecsdr_amd/libpss_$v.so python tools/time_fwd.py 2; fi
done 2>&1 | grep k_nfm
