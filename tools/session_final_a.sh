#!/bin/bash
# round artefacts, part A: GPU tests, smoke, default bench line, bound analysis, views table, cfg 4 on one GPU, fusion
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-final_a}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -15 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
echo "bench done"
tools/bench_variants.sh real noload addrmask real noload addrmask > $O/bound.txt 2>&1; cat $O/bound.txt
BENCH_ARGS=--float-images tools/bench_variants.sh real noload addrmask >> $O/bound.txt 2>&1
python tools/bench_views.py 2>/dev/null | grep -v "^{" > $O/views.txt; cat $O/views.txt
python bench.py --workload cfg4 > $O/cfg4.json 2> $O/cfg4.err; tail -c 600 $O/cfg4.json
python tools/bench_fusion.py > $O/fusion.json 2> $O/fusion.err; tail -c 400 $O/fusion.json
python tools/bench_scales.py > $O/scales.txt 2>&1; tail -5 $O/scales.txt
