#!/bin/bash
# split-bf16 mode: one-process plan sweep (scripts/gpu_x3_plansweep.py) and the row-threshold bench lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/x3; export TMPDIR=/tmp
O=gpurun_out/x3
timeout 400 python scripts/gpu_x3_plansweep.py ${FRAMES:-300} $O/plansweep_n${FRAMES:-300}.txt 2> $O/plansweep.err; echo "sweep exit $?"; tail -3 $O/plansweep.err
for m in 1 32 64 128; do
  RY_X3_MINM=$m timeout 200 python bench.py --no-cpu-baseline --dtype bf16x3 --steps 30 > $O/bench_x3_minm$m.json 2> $O/bench_x3_minm$m.err
  python -c "
import json; d = json.loads(open('$O/bench_x3_minm$m.json').read().strip().splitlines()[-1]); print('minM $m', d['value'], d['ms_per_step'], d['graph_replay_ms'])"
done
