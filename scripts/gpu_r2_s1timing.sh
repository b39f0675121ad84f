#!/bin/bash
# per-phase shader-clock stamps of every stage-1 layer (RY_S1_TIMING=1: eager launches, a sync and a read-back after each)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
RY_S1_TIMING=${RY_S1_TIMING:-1} RY_GRAPH=0 python - <<'PY' 2>&1 | grep -E "S1TIMING|Error|error" | tail -40
import sys
sys.path.insert(0, '.')
import torch
from realtime_yukarin_amd import engine, synth
from realtime_yukarin_amd.weights import flatten_params
ctx = engine.get_context(0)
(d1, P1), _ = synth.model_params('SYN-64')
net1 = engine.Net(ctx, d1, flatten_params(d1, P1))
net1.profile(1, 384, 1)
print('-----', file=sys.stderr)
net1.profile(1, 384, 1)
PY
