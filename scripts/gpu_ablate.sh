#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --profile-only --profile-reps 10"
for i in 1 2 3; do
echo "default (2-barrier ILV): $($B 2>/dev/null)"
echo "producer/consumer      : $(RY_PC=1 $B 2>/dev/null)"
done
RY_PC=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "syn64 or conv2d" 2>&1 | tail -2
