#!/bin/bash
# split-bf16 mode: one-process plan sweeps (scripts/gpu_x3_plansweep.py) at several window sizes, then the bench lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/x3; export TMPDIR=/tmp
O=gpurun_out/x3
for F in ${FRAMES:-300 100 400 1000}; do
  timeout 300 python scripts/gpu_x3_plansweep.py $F $O/plansweep_n$F.txt 2> $O/plansweep_$F.err | grep -v "^# split"; echo "sweep $F exit $?"
done
b() { name=$1; shift; timeout 200 python bench.py --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err; python -c "
import json; d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['graph_replay_ms'], d['roofline']['kernel'], d['roofline']['achieved'])"; }
b bench_x3_n300_v2 --dtype bf16x3 --steps 100 --layers-out $O/layers_x3_n300_v2.txt
b bench_x3_n100_v2 --dtype bf16x3 --frames 100 --steps 100
b bench_x3_n400_v2 --dtype bf16x3 --frames 400 --steps 100
b bench_x3_n1000_v2 --dtype bf16x3 --frames 1000 --steps 50
b bench_bf16_n400_v2 --dtype bf16 --frames 400 --steps 100
b bench_x3_n300_minm128_v2 --dtype bf16x3 --steps 100
timeout 300 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
