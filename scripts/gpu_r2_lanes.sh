#!/bin/bash
# lanes x stagger cut of the window call, interleaved on one box: gpu_r2_lanes.sh reps "ENV ..." ...
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R=$1; shift
for i in $(seq 1 $R); do for E in "$@"; do
  env $E timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 6 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[%-46s] step %.4f ms  s2 alone %.4f' % ('$E', d['ms_per_step'], d['graph_replay_ms']['stage2_alone']))"
done; done
