#!/bin/bash
# PMC comparison of the photometric and geometric-consistency update kernels (run on the GPU box via gpurun)
set -o pipefail
OUT=$PWD/gpurun_out/prof_geom
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:?}
python3 tools/bench_scales.py > $OUT/plain.json 2>/dev/null
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/pmc -- python3 tools/bench_scales.py > /dev/null 2> $OUT/pmc.err
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUT", "gpurun_out/prof_geom")
f = glob.glob(out + "/pmc/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "k_update" not in k: continue
    name = "geom" if "k_update<true, false" in k else ("prior" if "k_update<false, true" in k else "photo")
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, d in acc.items():
    print(name, {c: round(sum(v) / len(v) / 1e6, 1) for c, v in sorted(d.items())}, "launches", len(next(iter(d.values()))))
PY
