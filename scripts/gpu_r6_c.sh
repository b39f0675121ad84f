#!/bin/bash
# round 6, call c: the K loop with the filter fragments two positions ahead; plan sweep ranked by the graph-replayed forward
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_c; mkdir -p $O
SWEEP_SPLITS=${SWEEP_SPLITS:-0,1,2,3,4,5,6} timeout 2000 python scripts/gpu_r6_wino.py 300 $O/wino_sweep_n300.txt 6 > $O/sweep.log 2>&1; echo "sweep exit $?"; grep "^#" $O/wino_sweep_n300.txt | tail -40
