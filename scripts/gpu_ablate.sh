#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --profile-only --profile-reps 10"
for i in 1 2; do
echo "two-barrier ILV : $(RY_PIPE=0 $B 2>/dev/null)"
echo "pipelined       : $($B 2>/dev/null)"
echo "pipelined t256  : $(RY_TILE64=256 $B 2>/dev/null)"
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
