#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --profile-only --profile-reps 10"
for i in 1 2 3; do
echo "depth 1 (default): $($B 2>/dev/null)"
echo "depth 2          : $(RY_DEEP=1 $B 2>/dev/null)"
echo "depth 2, no ILV  : $(RY_DEEP=1 RY_ILV=0 $B 2>/dev/null)"
done
RY_DEEP=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "syn64 or conv2d" 2>&1 | tail -2
