set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
python -m pytest tests/test_literal_gpu.py -x -q -s > gpurun_out/r4f/literal.log 2>&1; echo "literal rc=$?"; grep -E "T2|max \||within 1|HIP vs|passed|failed" gpurun_out/r4f/literal.log | tail -12
python -m pytest tests/test_fullsize_gpu.py -x -q --durations=5 > gpurun_out/r4f/fullsize.log 2>&1; echo "fullsize rc=$?"; tail -12 gpurun_out/r4f/fullsize.log
