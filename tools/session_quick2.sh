#!/bin/bash
# quick check: parity subset, cfg 1 bench twice, fp32, views table
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-quick}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py -x -q > $O/pytest.log 2>&1 || { tail -15 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
tools/bench_variants.sh real real 2>&1 | tee $O/variants.txt
BENCH_ARGS=--float-images tools/bench_variants.sh real 2>&1 | tee -a $O/variants.txt
VIEWS=${VIEWS:-8,20} python tools/bench_views.py 2>/dev/null | grep -v "^{" | tee $O/views.txt
