#!/bin/bash
# stage-1 alone at 300 frames under RY_S1_UNITS = 128 .. 1024 and a few forced slices (interleaved, 3 rounds)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
run() { python - <<PY
import sys, numpy
sys.path.insert(0, '.')
import torch
from realtime_yukarin_amd import engine, synth
from realtime_yukarin_amd.weights import flatten_params
ctx = engine.get_context(0)
(d1, P1), _ = synth.model_params('SYN-64')
net1 = engine.Net(ctx, d1, flatten_params(d1, P1))
N = 300
dx = ctx.dev_alloc(N * 9); dy = ctx.dev_alloc(N * 9); ctx.dev_upload(dx, synth.stage1_input(N)[0])
for _ in range(5): net1.convert_device(dx, dy, 1, N)
ctx.sync(); best = 1e9
for rep in range(5):
    ctx.timer_start()
    for _ in range(50): net1.convert_device(dx, dy, 1, N)
    best = min(best, ctx.timer_stop() / 50)
print('%-40s %.4f ms' % ('$1', best))
PY
}
for rep in 1 2 3; do
  for U in 128 256 512 1024; do RY_S1_UNITS=$U run "units=$U"; done
  RY_S1_CFG="9:2:8,10:2:8,11:2:8,12:2:8,13:2:8,14:2:8" run "decoder layers at 2x8"
  RY_S1_CFG="2:4:4,3:4:4,4:4:4,5:4:4" run "encoder c2-c5 at 4x4"
done
