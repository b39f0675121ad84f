#!/bin/bash
# round 6, call e: plan candidates for single layers with every other layer on the Winograd defaults, in turn on one box
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_e; mkdir -p $O
timeout 1500 python scripts/gpu_r6_plan_ab.py 300 4 $O/plan_ab_n300.txt - "env:RY_WINO=12:1:2:3" "env:RY_WINO=13:1:2:3" "env:RY_WINO=12:1:2:3,13:1:2:3" "env:RY_WINO=1:2:1:1" "env:RY_WINO=3:2:4:5" "env:RY_WINO=2:1:1:4" "env:RY_WINOGRAD=0" > $O/ab.log 2>&1; echo "ab exit $?"; grep "^#" $O/plan_ab_n300.txt | cut -c1-150
