#!/bin/bash
# Round 4, call D: the final bench.py -- the driver's command in 10 fresh processes, then the default run (all extras).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_d; mkdir -p $O
bash scripts/gpu_r4_driver_cmd.sh 10 > $O/driver_cmd.log 2>&1; tail -3 $O/driver_cmd.log
cp gpurun_out/r04_driver_cmd/summary.txt $O/driver_cmd_summary.txt
timeout 600 python3 bench.py --layers-out $O/layers.txt --details-out $O/details.json > $O/bench_default.json 2> $O/bench_default.err; echo "default exit $?"; wc -c $O/bench_default.json; cat $O/bench_default.json
