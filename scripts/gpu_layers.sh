#!/bin/bash
# per-layer tables of the stage-2 predictor for kernel variants (RY_LDSDMA values given as arguments)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "$@"; do
RY_LDSDMA=$v python bench.py --profile-only --profile-reps 10 --layers-out gpurun_out/layers_dma$v.txt > /dev/null 2>&1
echo "== RY_LDSDMA=$v"; cat gpurun_out/layers_dma$v.txt
done
