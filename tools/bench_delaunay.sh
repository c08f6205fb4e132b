#!/bin/bash
# host Delaunay on the GPU box's CPUs: thread scaling, with the box's CPU facts (affinity, cgroup quota)
cd ${GRAFT_REPO_ROOT:-.}
g++ -O2 -std=c++17 -fopenmp -I mp-mvs_amd/host -I include tools/bench_delaunay.cpp mp-mvs_amd/host/planar_prior.cpp -o build/bench_delaunay -lpthread -Lmp-mvs_amd/csrc -lmpmvs_hip -Wl,-rpath,$PWD/mp-mvs_amd/csrc || exit 1
nproc; lscpu | grep -E "Model name|Thread|Core|Socket|NUMA node\(s\)"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; taskset -p $$ | tail -1
for t in 1 2 4 8 16 32; do echo "threads $t"; MPMVS_HOST_TIMING=1 MPMVS_HOST_THREADS=$t build/bench_delaunay 2>&1 | grep -E "build|tris" | sort -k4 -n | awk 'NR<=2 || /tris/' | tail -4; done
