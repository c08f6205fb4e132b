#!/bin/bash
# full GPU test suite, smoke, default bench line -> gpurun_out/<tag>/
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-final}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -15 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - "$O" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + '/bench.json').read().strip().splitlines()[-1])
print(d['value'], 'Mpix/s', d['roofline']['avg_launch_ms'], 'ms frac', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'])
for k, v in d['secondary'].items():
    print(k, {a: b for a, b in v.items() if a not in ('workload',)})
PY
