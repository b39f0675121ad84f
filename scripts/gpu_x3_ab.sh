#!/bin/bash
# same-box A/B of the split-bf16 planner's main-loop rate (RY_PLAN_X3_PEAK, TFLOP/s of bf16 products) after the issue-placement change
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/x3
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-split-bf16 --dtype bf16x3 --steps 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'], d['graph_replay_ms']['stage2_alone'])" | tee -a gpurun_out/x3/ab_peak.txt; }
run peak1150 RY_PLAN_X3_PEAK=1150
run peak1500 RY_PLAN_X3_PEAK=1500
run peak2000 RY_PLAN_X3_PEAK=2000
run peak1150 RY_PLAN_X3_PEAK=1150
run peak1500 RY_PLAN_X3_PEAK=1500
run peak2000 RY_PLAN_X3_PEAK=2000
run dec_c4_s2 RY_PLAN=12:1:2:1
