#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for f in 0 4 8 12; do
RY_IGEMM_DBG=$f python bench.py --profile-only --profile-reps 10 --layers-out gpurun_out/l_$f.txt >/dev/null 2>&1
echo "== dbg $f"; grep -v "splitk\|sr_" gpurun_out/l_$f.txt
done
