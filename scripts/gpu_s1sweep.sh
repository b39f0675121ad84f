#!/bin/bash
# stage-1 split heuristic sweep (tuning aid): graph-replay time of the stage-1 predictor alone
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "128 16" "256 16" "256 32" "512 32" "512 64" "1024 64"; do
  set -- $cfg
  RY_S1_WGS=$1 RY_S1_MAXS=$2 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('wgs/maxs $cfg:', d['graph_replay_ms'], 'ms/step', d['ms_per_step'], 'stage1 kernels ms', d['roofline_stage1']['kernel_ms_per_forward'])
"
done
