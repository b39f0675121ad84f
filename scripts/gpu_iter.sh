#!/bin/bash
# quick iteration: stage-2 parity, kernel-family profile, layer table, default bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
B="python bench.py --profile-only --profile-reps 10"
for i in 1 2; do echo "profile: $($B 2>/dev/null)"; done
bash scripts/gpu_layers.sh 1 | grep -v "^=="
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_iter.json; cat gpurun_out/bench_iter.json
