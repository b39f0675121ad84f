#!/bin/bash
# bench at the BASELINE window sizes and a few batch sizes (numbers for DESIGN.md section 6)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for cfg in "--frames 100" "--frames 300" "--frames 400" "--frames 600" "--frames 1000" "--frames 300 --windows 4" "--frames 300 --windows 8" "--frames 1000 --windows 8" "--frames 400 --dtype bf16" "--frames 1000 --dtype bf16" "--frames 400 --dtype bf16 --windows 8"; do
  python bench.py $cfg --no-cpu-baseline --steps 30 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('%-44s value %9.0f  x_rt %6.0f  ms/step %7.3f  s1 %.3f s2 %.3f  | %s %.1f TF frac %.3f | host_call %.3f ms' % ('$cfg', d['value'], d['x_realtime'], d['ms_per_step'], d['graph_replay_ms']['stage1_alone'], d['graph_replay_ms']['stage2_alone'], r['kernel'], r['achieved'], r['frac'], d['host_call_ms_per_window']))
"
done
