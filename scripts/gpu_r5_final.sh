#!/bin/bash
# Round 5, the final GPU call: the whole GPU suite, smoke(), then the evidence set of the final source (scripts/gpu_r5_profile.sh r05_z).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05_z; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
bash scripts/gpu_r5_profile.sh r05_z
