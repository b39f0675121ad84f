#!/bin/bash
# quick iteration: GPU tests, kernel-family profile, default bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
python bench.py --no-cpu-baseline --layers-out gpurun_out/layers_iter.txt 2>/dev/null | tail -1 > gpurun_out/bench_iter.json
python -c "
import json; d=json.loads(open('gpurun_out/bench_iter.json').read())
print('value', d['value'], 'ms/step', d['ms_per_step'], d['graph_replay_ms'])
print('roofline', d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['frac'], 'traffic', d['roofline']['traffic'])
print('stage1', d['roofline_stage1'])
"
grep stage1 gpurun_out/layers_iter.txt
