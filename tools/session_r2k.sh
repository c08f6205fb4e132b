#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r2k; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -15 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
tools/bench_variants.sh idxen real idxen real 2>&1 | tee $O/variants.txt
VIEWS=8,16,20,24 python tools/bench_views.py > $O/views.txt 2>&1; grep "V=" $O/views.txt
