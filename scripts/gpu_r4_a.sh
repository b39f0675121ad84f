#!/bin/bash
# Round 4, call A: the driver's command with the collector fix (6 fresh processes), the full GPU suite, the dispatcher measurement.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_a; mkdir -p $O
bash scripts/gpu_r4_driver_cmd.sh 6 > $O/driver_cmd.log 2>&1; tail -4 $O/driver_cmd.log
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.txt
timeout 600 python bench.py --dispatcher > $O/dispatcher.json 2> $O/dispatcher.err; echo "dispatcher exit $?"; cat $O/dispatcher.json; tail -3 $O/dispatcher.err
timeout 600 python bench.py --dispatcher --dispatcher-depth 3 > $O/dispatcher_d3.json 2> $O/dispatcher_d3.err; echo "dispatcher d3 exit $?"; cat $O/dispatcher_d3.json
nproc; lscpu | grep "Model name"
