cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "compaction_select" 2>&1 | tail -15
PSS_LIBRARY=pyspecsdr_amd/libpss_nocompact.so timeout 900 python -m pytest tests -m gpu -x -q -k "compaction_select" 2>&1 | tail -5
