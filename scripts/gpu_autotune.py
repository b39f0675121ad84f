#!/usr/bin/env python3
"""RY_AUTOTUNE on the GPU: time to tune a plan, the tuned against the planner's plans (stage-2 graph replay, fp32 and split-bf16),
and that the results agree.  Usage (GPU box): python scripts/gpu_autotune.py [frames]"""
import ctypes
import os
import sys
import time
from pathlib import Path

import numpy
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from realtime_yukarin_amd import engine, synth                      # noqa: E402
from realtime_yukarin_amd.weights import flatten_params             # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.cuda.init()                                                    # like bench.py: torch's HIP context first, then the library's
(_, _), (d2, P2) = synth.model_params('SYN-64')
ctx = engine.get_context(0)
net = engine.Net(ctx, d2, flatten_params(d2, P2), width=synth.FFT_BINS - 1)
sp = synth.stage2_input(N)[0]


def reread():
    ctx.reload_env()


def replay_ms(reps=40):
    x = torch.from_numpy(sp[None]).cuda(); y = torch.empty_like(x)
    for _ in range(3):
        net.convert_device(x.data_ptr(), y.data_ptr(), 1, N)
    ctx.sync(); ctx.timer_start()
    for _ in range(reps):
        net.convert_device(x.data_ptr(), y.data_ptr(), 1, N)
    return ctx.timer_stop() / reps


out = {}
for tune in ('0', '1'):
    os.environ['RY_AUTOTUNE'] = tune
    reread()
    for mode in ('f32', 'bf16x3'):
        net.set_dtype(mode)
        t = time.perf_counter(); y = net.convert(sp); dt = time.perf_counter() - t
        out[(tune, mode)] = y
        print('autotune=%s %-6s first convert (plan build) %.2f s, stage-2 replay %.4f ms' % (tune, mode, dt, replay_ms()), flush=True)
for mode in ('f32', 'bf16x3'):
    a, b = out[('0', mode)], out[('1', mode)]
    print(mode, 'tuned vs planner plans: max rel diff %.2e' % float(numpy.abs(a / b - 1).max()))
net.close()
