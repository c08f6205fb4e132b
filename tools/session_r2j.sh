#!/bin/bash
# one GPU session: idxen variant A/B + parity of the variant, cfg 3 with the parallel Delaunay, host thread scaling
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r2j; mkdir -p $O
nproc > $O/host.txt; lscpu | grep -i "model name" >> $O/host.txt
g++ -O2 -std=c++17 -fopenmp -I mp-mvs_amd/host -I include tools/bench_delaunay.cpp mp-mvs_amd/host/planar_prior.cpp -o build/bench_delaunay -lpthread 2>> $O/host.txt
for t in 1 2 4 8 16 32; do echo "threads $t" >> $O/host.txt; MPMVS_HOST_THREADS=$t build/bench_delaunay 2>&1 | tail -2 >> $O/host.txt; done
MPMVS_HIP_LIB=$PWD/build/libmpmvs_hip_idxen.so timeout -k 10 500 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py -x -q > $O/pytest_idxen.log 2>&1 || { tail -5 $O/pytest_idxen.log; exit 1; }
tail -2 $O/pytest_idxen.log
tools/bench_variants.sh real idxen real idxen 2>&1 | tee $O/variants.txt
BENCH_ARGS=--float-images tools/bench_variants.sh real idxen 2>&1 | tee -a $O/variants.txt
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2j/bench.json').read().strip().splitlines()[-1])
print(json.dumps(d['secondary']['cfg3'])); print(d['value'], d['roofline']['avg_launch_ms'])
PY
