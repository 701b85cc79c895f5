cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/cumask_probe.py 2>&1 | tail -12
