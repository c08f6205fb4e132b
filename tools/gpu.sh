#!/bin/bash
# Local wrapper around gpurun: stamps the tree's git head into build/git_head.txt (the snapshot that travels to the GPU box has
# no .git; tools/summarize_prof.py copies the stamp into the profile summaries) and forwards everything to gpurun.
#   tools/gpu.sh --timeout 900 -- 'python -m pytest tests -m gpu -x -q'
cd "$(dirname "$0")/.." || exit 1
mkdir -p build
git describe --always --dirty > build/git_head.txt
exec /usr/local/graft/bin/gpurun "$@"
