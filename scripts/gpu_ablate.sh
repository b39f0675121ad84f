#!/bin/bash
# ablation of the LDS-DMA kernel (RY_IGEMM_DBG: 4 no output stores, 8 no K loop, 128 no loads in the K loop) -- wrong results, timing only
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --profile-only --profile-reps 10"
for i in 1 2; do
for f in 0 128 132; do echo "dbg $f: $(RY_IGEMM_DBG=$f $B 2>/dev/null | cut -c1-260)"; done
done
