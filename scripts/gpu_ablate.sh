#!/bin/bash
# A/B of the stage-2 implicit-GEMM variants: K groups on/off, register-staged kernel
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "syn64" 2>&1 | tail -3
B="python bench.py --profile-only --profile-reps 10"
for i in 1 2 3; do
echo "LDS-DMA + K groups     : $($B 2>/dev/null)"
echo "LDS-DMA, no K groups   : $(RY_KGROUPS=0 $B 2>/dev/null)"
done
bash scripts/gpu_layers.sh 1
