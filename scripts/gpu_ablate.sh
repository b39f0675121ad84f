#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --profile-only --profile-reps 10 --layers-out gpurun_out/l.txt"
for i in 1 2; do
echo "default          : $($B 2>/dev/null)"
echo "256x128 big tile : $(RY_BIGTILE=1 $B 2>/dev/null)"; grep "igemm_f32<256" gpurun_out/l.txt
done
