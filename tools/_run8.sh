cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_full.txt 2>&1
tail -6 $O/pytest_full.txt
