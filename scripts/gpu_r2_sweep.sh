#!/bin/bash
# Window-size sweep of the chained step + the config-#1 schedule on the GPU.  Output: gpurun_out/sweep/
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/sweep; mkdir -p $O; export TMPDIR=/tmp
S='--no-cpu-baseline --no-extras --steps 40'
for F in 100 200 300 400 600 1000; do
  EX=0; [ $F = 300 ] && EX=100; [ $F = 600 ] && EX=200; [ $F = 400 ] && EX=100
  timeout 200 python bench.py $S --frames $F --extra-frames $EX > $O/n$F.json 2> $O/n$F.err
  python - <<PY
import json; d=json.load(open('$O/n$F.json'))
print('frames $F: %.0f frames/s  %.4f ms/step  x_rt %.0f  effective %.0f  dominant %s %.1f TF  stage2 fwd frac %.3f  stage1 %.4f ms' % (d['value'], d['ms_per_step'], d['x_realtime'], d['effective_x_realtime'], d['roofline']['kernel'], d['roofline']['achieved'], d['roofline_stage2_forward']['frac'], d['graph_replay_ms']['stage1_alone']))
PY
done
for W in 4 8; do
  timeout 200 python bench.py $S --windows $W > $O/w$W.json 2> $O/w$W.err
  python - <<PY
import json; d=json.load(open('$O/w$W.json')); print('windows $W x 300: %.0f frames/s %.4f ms/step' % (d['value'], d['ms_per_step']))
PY
done
timeout 200 python bench.py $S --dtype bf16 --frames 400 --extra-frames 100 > $O/bf16_n400.json 2> $O/bf16.err
python - <<PY
import json; d=json.load(open('$O/bf16_n400.json')); print('bf16 400 frames: %.0f frames/s %.4f ms/step' % (d['value'], d['ms_per_step']))
PY
