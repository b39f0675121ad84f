#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --profile-only --profile-reps 10"
for i in 1 2 3; do
echo "default          : $($B 2>/dev/null)"
echo "stagger prio     : $(RY_STAGGER=1 $B 2>/dev/null)"
done
