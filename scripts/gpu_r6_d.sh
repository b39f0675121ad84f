#!/bin/bash
# round 6, call d: what clock does the chip sustain under the direct kernels, the Winograd defaults and the per-layer winners?
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_d; mkdir -p $O
timeout 600 python scripts/gpu_r6_clocks.py 300 $O/clocks_n300.txt "1:2:1:1,2:1:1:4,3:2:4:5,4:1:2:0,11:1:2:0,12:1:2:3,13:1:2:3,14:1:2:0" "12:1:2:3,13:1:2:3" "12:1:2:3" > $O/clocks.log 2>&1; echo "clocks exit $?"; cat $O/clocks_n300.txt; tail -5 $O/clocks.log
