#!/bin/bash
# Round 5, GPU call H: the cooperative column minimum of the fused stage-1 pad (per-layer table, stage-1 forward at 300 / 100 / 1000 frames) and the
# layer table of config #5 (bf16 stage 2, 400 frames).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_h; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "stage1 or ac_convert or predictor or conv1d or golden" > $O/pytest_s1.txt 2>&1; echo "pytest s1 exit $?"; tail -3 $O/pytest_s1.txt
timeout 400 python bench.py --no-cpu-baseline --layers-out $O/layers_f32.txt --details-out $O/details_f32.json > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench f32 exit $?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r5_h/bench_f32.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'], 'graph_replay', d.get('graph_replay_ms'), 'roofline_stage1', d.get('roofline_stage1'))
PY
grep stage1 $O/layers_f32.txt
timeout 400 python bench.py --no-cpu-baseline --no-extras --frames 400 --dtype bf16 --layers-out $O/layers_bf16_n400.txt --details-out $O/details_bf16.json > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 exit $?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r5_h/bench_bf16.json'))
print('bf16 n400: value', d['value'], 'ms/step', d['ms_per_step'], 'graph_replay', d.get('graph_replay_ms'))
PY
grep stage2 $O/layers_bf16_n400.txt
