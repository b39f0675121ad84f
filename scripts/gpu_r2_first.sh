#!/bin/bash
# Round-2 first GPU call: baseline of HEAD (GPU suite, autotuner vs planner, default bench line).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r2a; mkdir -p $O; export TMPDIR=/tmp
timeout 500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest gpu exit $?"; tail -3 $O/pytest_gpu.txt
timeout 300 python scripts/gpu_autotune.py 300 > $O/autotune_n300.txt 2>&1; echo "autotune exit $?"; tail -12 $O/autotune_n300.txt
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench exit $?"; head -c 1500 $O/bench_default.json; echo
nproc; lscpu | grep "Model name"
