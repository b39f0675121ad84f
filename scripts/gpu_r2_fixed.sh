#!/bin/bash
# What a stage-2 implicit-GEMM launch costs besides its K loop: RY_IGEMM_DBG=8 skips the K loop (prologue + epilogue + stores remain),
# 12 also skips the output stores, 4 skips only the stores.  Diagnostics with wrong results; per-layer eager times + graph replay.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/fixed; mkdir -p $O; export TMPDIR=/tmp
for D in 0 8 12 4; do
  RY_IGEMM_DBG=$D timeout 200 python - <<PY > $O/dbg$D.txt 2>&1
import sys, numpy
sys.path.insert(0, '.')
import torch
from realtime_yukarin_amd import engine, synth
from realtime_yukarin_amd.weights import flatten_params
ctx = engine.get_context(0)
(d1, P1), (d2, P2) = synth.model_params('SYN-64')
net2 = engine.Net(ctx, d2, flatten_params(d2, P2), width=512)
T = 384
din = ctx.dev_alloc(T * 512); dout = ctx.dev_alloc(T * 512)
ctx.dev_upload(din, numpy.random.default_rng(0).normal(size=T * 512).astype('f4'))
for _ in range(5): net2.forward_device(din, dout, 1, T)
ctx.sync(); ctx.timer_start()
for _ in range(30): net2.forward_device(din, dout, 1, T)
print('graph replay ms', ctx.timer_stop() / 30)
for q in net2.profile(1, T, 5):
    print('%-12s %-40s %8.2f us' % (q['layer'], q['name'], q['ms'] * 1e3))
PY
  echo "== RY_IGEMM_DBG=$D"; head -1 $O/dbg$D.txt; grep igemm $O/dbg$D.txt | awk '{printf "%s %s %s | ", $1, $3, $4} END {print ""}'
done
