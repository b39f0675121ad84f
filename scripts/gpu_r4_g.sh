#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke exit $?"; tail -6 $O/smoke.txt
