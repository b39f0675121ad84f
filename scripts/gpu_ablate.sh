#!/bin/bash
# A/B of the stage-2 implicit-GEMM kernels: LDS-DMA (default) vs register-staged (RY_LDSDMA=0)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --profile-only --profile-reps 10"
for i in 1 2 3; do
echo "LDS-DMA                : $($B 2>/dev/null)"
echo "register-staged        : $(RY_LDSDMA=0 $B 2>/dev/null)"
done
