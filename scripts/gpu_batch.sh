#!/bin/bash
# bench at a few batch / frame configurations (stage-1 roofline in the batched regime)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for cfg in "--frames 300 --windows 1" "--frames 1000 --windows 1" "--frames 1000 --windows 8" "--frames 300 --windows 8"; do
  echo "== $cfg"
  python bench.py $cfg --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'])
print('roofline', {k: d['roofline'][k] for k in ('kernel','achieved','frac') if k in d['roofline']})
print('stage1', d.get('roofline_stage1'))
"
done
