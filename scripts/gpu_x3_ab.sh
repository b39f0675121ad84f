#!/bin/bash
# same-box A/B inside the split-bf16 mode (interleaved): last layer on the [hi | lo] copies (default) vs on fp32 copies (RY_X3_LAST=0)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/x3
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -s -k "x3_variant or last" 2>&1 | tail -3
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --dtype bf16x3 --steps 100 --layers-out gpurun_out/x3/layers_$tag.txt 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'], d['graph_replay_ms']['stage2_alone'], {k: v['ms'] for k, v in d['kernels'].items() if 'sr_' in k})"; }
for i in 1 2; do
run last_fp32 RY_X3_LAST=0
run last_x3 RY_X3_LAST=1
done
run f32 RY_X3_LAST=1 2>/dev/null
python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32 mode', d['value'], d['ms_per_step'], d['graph_replay_ms']['stage2_alone'], {k: v['ms'] for k, v in d['kernels'].items() if 'sr_' in k})"
