#!/bin/bash
# The final GPU call of a round: the whole GPU suite, smoke(), then the evidence set of the final source (scripts/gpu_profile.sh <round> <tag>).
#   gpu_final.sh [round] [tag]  ->  gpurun_out/<round>_<tag>/...
cd "$GRAFT_REPO_ROOT"; RND=${1:-r06}; TAG=${2:-z}; O=gpurun_out/${RND}_$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -6 | tee $O/smoke.txt
bash scripts/gpu_profile.sh $RND $TAG
