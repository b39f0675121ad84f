#!/bin/bash
# interleaved A/B of any number of environment settings on one box: gpu_r2_ab3.sh reps "VAR=.. VAR=.." "VAR=.." ...
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/ab3; mkdir -p $O; export TMPDIR=/tmp
R=$1; shift
for i in $(seq 1 $R); do k=0; for E in "$@"; do k=$((k+1))
  env $E timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 5 > $O/c${k}_$i.json 2> $O/c${k}_$i.err
  python - <<PY
import json; d=json.load(open('$O/c${k}_$i.json'))
print('[%-34s] step %.4f ms  %s' % ('$E', d['ms_per_step'], d['graph_replay_ms']))
PY
done; done
