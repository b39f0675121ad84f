#!/bin/bash
# Round 4, call F: the evidence set of the final tree -- rocprofv3 summaries + PMC passes + bench line (gpu_r4_profile.sh), the full GPU suite, both soaks, the dispatcher measurement.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_f; mkdir -p $O
bash scripts/gpu_r4_profile.sh r04_p > $O/profile.log 2>&1; tail -12 $O/profile.log
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.txt
timeout 300 python scripts/gpu_dispatch_soak.py 3000 2 > $O/dispatch_soak.txt 2>&1; echo "dispatch soak exit $?"; tail -3 $O/dispatch_soak.txt
timeout 300 python scripts/gpu_soak.py 3000 > $O/soak.txt 2>&1; echo "soak exit $?"; tail -3 $O/soak.txt
timeout 600 python bench.py --dispatcher > $O/dispatcher.json 2> $O/dispatcher.err; echo "dispatcher exit $?"; cat $O/dispatcher.json
