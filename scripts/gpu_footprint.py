#!/usr/bin/env python3
"""HBM footprint of the predictors and of a window core (round-5 advisor: say what the extra filter layouts cost): device-wide free memory
(hipMemGetInfo through torch) before and after each step, SYN-64, 300-frame windows.    usage (GPU box): python scripts/gpu_footprint.py [frames]"""
import os
import sys
from pathlib import Path

import numpy
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
from realtime_yukarin_amd import engine, sptk, synth                # noqa: E402
from realtime_yukarin_amd.weights import flatten_params             # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.cuda.init()
free = lambda: torch.cuda.mem_get_info()[0] / 2**20
(d1, P1), (d2, P2) = synth.model_params('SYN-64')
ctx = engine.get_context(0)
f0 = free()
w1, w2 = flatten_params(d1, P1), flatten_params(d2, P2)
n1 = engine.Net(ctx, d1, w1); ctx.sync(); f1 = free()
n2 = engine.Net(ctx, d2, w2, width=synth.FFT_BINS - 1); ctx.sync(); f2 = free()
print('stage-1 predictor: %.1f MB of parameters -> %.0f MB of HBM (two filter layouts)' % (w1.nbytes / 2**20, f0 - f1))
print('stage-2 predictor: %.1f MB of parameters -> %.0f MB of HBM at creation (implicit-GEMM + direct layouts, ry_c2d_os layout of the weight-streaming layers)' % (w2.nbytes / 2**20, f1 - f2))
sp = synth.stage2_input(N)[0]
n2.convert(sp); ctx.sync(); f3 = free()
print('first %d-frame stage-2 window: +%.0f MB (Winograd filters of the layers that take that path, built on first use; the plan: activation buffers, slabs, graph)' % (N, f2 - f3))
core = engine.VcCore(n1, n2, sptk.mc2sp_matrix(8, sptk.mcepalpha(16000), 1024), lanes=2)
x = synth.stage1_input(N)[0]
eff = numpy.ones(N, bool)
for _ in range(4):
    core.convert(x, eff)
ctx.sync(); f4 = free()
print('window core with two lanes (clones: own plans / activations / graphs, shared filters; six pinned ring slots): +%.0f MB' % (f3 - f4))
print('total %.0f MB of 288 GB' % (f0 - f4))
