#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_dbg; mkdir -p $O
for i in 1 2 3; do
  BENCH_DEBUG_FENCE=1 timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --details-out $O/d_$i.json > $O/run_$i.json 2> $O/run_$i.err; echo "run $i exit $?"; grep -h "slowest\|ms for the steps" $O/run_$i.err | head -20
done
