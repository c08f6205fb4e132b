cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_pipeline_gpu.py -x -q -k "rccl" 2>&1 | tail -8
