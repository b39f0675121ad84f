#!/bin/bash
# bf16 stage-2 (BASELINE config #5): parity + bench at 400 frames, fp32 beside it
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k bf16 2>&1 | tail -3
for dt in bf16 f32; do
python bench.py --no-cpu-baseline --dtype $dt --frames 400 --layers-out gpurun_out/layers_$dt.txt 2>/dev/null | tail -1 > gpurun_out/bench_n400_$dt.json
python -c "
import json; d=json.loads(open('gpurun_out/bench_n400_$dt.json').read())
print('$dt value', d['value'], 'ms/step', d['ms_per_step'], d['graph_replay_ms'])
print('roofline', d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['frac'])
"
done
grep stage2 gpurun_out/layers_bf16.txt
