#!/bin/bash
# A/B of the stage-2 implicit-GEMM kernels: LDS-DMA (default) vs filters-in-registers vs register-staged (RY_LDSDMA=0)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "syn64 or conv2d" 2>&1 | tail -3
B="python bench.py --profile-only --profile-reps 10"
for i in 1 2 3; do
echo "LDS-DMA                : $($B 2>/dev/null)"
echo "LDS-DMA, B direct      : $(RY_BDIRECT=1 $B 2>/dev/null)"
echo "register-staged        : $(RY_LDSDMA=0 $B 2>/dev/null)"
done
bash scripts/gpu_layers.sh 1
