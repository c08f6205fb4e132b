cd $GRAFT_REPO_ROOT
python -m pytest tests/test_q8_gpu.py -x -q -s -k T1 2>&1 | grep -E "assert|Error|cost - literal|where" | head -12
