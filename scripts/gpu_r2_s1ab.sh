#!/bin/bash
# stage-1 A/B on one box: current library with / without the fused pad, and an older build (realtime_yukarin_amd/libry355_<tag>.so.ab)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; L=realtime_yukarin_amd/libry355.so; cp $L /tmp/cur.so; cp realtime_yukarin_amd/libry355_$1.so.ab /tmp/old.so
run() { python - <<PY
import sys, numpy
sys.path.insert(0, '.')
import torch
from realtime_yukarin_amd import engine, synth
from realtime_yukarin_amd.weights import flatten_params
ctx = engine.get_context(0)
(d1, P1), _ = synth.model_params('SYN-64')
net1 = engine.Net(ctx, d1, flatten_params(d1, P1))
res = []
for N in (100, 300, 1000):
    dx = ctx.dev_alloc(N * 9); dy = ctx.dev_alloc(N * 9); ctx.dev_upload(dx, synth.stage1_input(N)[0])
    for _ in range(5): net1.convert_device(dx, dy, 1, N)
    ctx.sync(); best = 1e9
    for rep in range(5):
        ctx.timer_start()
        for _ in range(50): net1.convert_device(dx, dy, 1, N)
        best = min(best, ctx.timer_stop() / 50)
    res.append('%d: %.4f ms' % (N, best))
print('$1', ' | '.join(res))
PY
}
for rep in 1 2 3; do
  cp /tmp/cur.so $L; run "cur padfuse=1"; RY_S1_PADFUSE=0 run "cur padfuse=0"
  cp /tmp/old.so $L; run "old"
done
cp /tmp/cur.so $L
