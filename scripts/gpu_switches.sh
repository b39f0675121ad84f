#!/bin/bash
# the A/B switches still produce correct results: GPU parity tests under each non-default setting
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "RY_LDSDMA=0" "RY_PATCH=0" "RY_PATCH=1" "RY_KGROUPS=0" "RY_GRAPH=0" "RY_TILE2D=0" "RY_S2_CROP=0" "RY_S2_CROP=1" "RY_VC_LANES=1" "RY_VC_LANES=3" "RY_XCD_GROUPS=0" "RY_S1_OS=0"; do
echo "== $v: $(env $v timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -1)"
done
