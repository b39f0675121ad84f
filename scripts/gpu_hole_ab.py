#!/usr/bin/env python3
"""Round 5: the encoder's identical padding rows copied instead of computed (RY_S2_HOLE=1, default) against computing them (=0), one process,
interleaved: stage-2 forward alone (graph replay) and the chained two-lane step exactly as bench.py's step.
usage (GPU box): python scripts/gpu_hole_ab.py [frames] [alternations] [out file]"""
import ctypes
import os
import sys
import time
from pathlib import Path

import numpy

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
from realtime_yukarin_amd import engine, sptk, synth                # noqa: E402
from realtime_yukarin_amd.weights import flatten_params             # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ALT = int(sys.argv[2]) if len(sys.argv) > 2 else 10
OUT = sys.argv[3] if len(sys.argv) > 3 else str(ROOT / 'gpurun_out' / ('r5_hole_ab_n%d.txt' % N))
(d1, P1), (d2, P2) = synth.model_params('SYN-64')
ctx = engine.get_context(0)
n1 = engine.Net(ctx, d1, flatten_params(d1, P1))
n2 = engine.Net(ctx, d2, flatten_params(d2, P2), width=synth.FFT_BINS - 1)
mtx = sptk.mc2sp_matrix(8, sptk.mcepalpha(16000), 1024)
x = synth.stage1_input(N)[0]
d_x = ctx.dev_alloc(N * 9); ctx.dev_upload(d_x, x)
d_rows = ctx.dev_alloc(N); ctx.dev_upload(d_rows, numpy.arange(N, dtype=numpy.int32))
d_mc = [ctx.dev_alloc(N * 9) for _ in range(6)]
d_sp = [ctx.dev_alloc(N * 513) for _ in range(6)]
d_in = ctx.dev_alloc(N * 513); d_out = ctx.dev_alloc(N * 513)
ctx.dev_upload(d_in, synth.stage2_input(N)[0])
reread = lambda: ctx.reload_env()
lines = []


def say(s):
    lines.append(s + '\n'); print(s, flush=True)


def setup(hole):
    os.environ['RY_S2_HOLE'] = hole; reread(); n2.set_dtype('f32')


def forward_alone(reps=40):
    for _ in range(3):
        n2.convert_device(d_in, d_out, 1, N)
    ctx.sync(); ctx.timer_start()
    for _ in range(reps):
        n2.convert_device(d_in, d_out, 1, N)
    return ctx.timer_stop() / reps


def two_lane(steps=100):
    core = engine.VcCore(n1, n2, mtx, lanes=2)
    k = [0]

    def step():
        core.enqueue_device(d_x, d_rows, N, N, d_mc[k[0] % 6], d_sp[k[0] % 6], 1e-16)
        k[0] += 1
    for _ in range(18):
        step()
    ctx.sync()
    best = 1e9
    for _ in range(3):
        for _ in range(4):
            step()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    core.close()
    return best


say('# RY_S2_HOLE=0 (identical padding rows of encoder c1 / c2 computed) against =1 (copied), SYN-64, %d frames, %d alternations' % (N, ALT))
rows = []
for r in range(ALT):
    setup('0'); f0 = forward_alone(); t0 = two_lane()
    setup('1'); f1 = forward_alone(); t1 = two_lane()
    rows.append((f0, f1, t0, t1))
    say('%2d  forward alone %.4f -> %.4f ms   two-lane step %.4f -> %.4f ms per window' % (r, f0, f1, t0, t1))
a = numpy.array(rows)
say('# median: forward alone %.4f -> %.4f ms; two-lane step %.4f -> %.4f ms (%+.2f %%); two-lane faster in %d of %d alternations'
    % (numpy.median(a[:, 0]), numpy.median(a[:, 1]), numpy.median(a[:, 2]), numpy.median(a[:, 3]),
       100 * (numpy.median(a[:, 3]) / numpy.median(a[:, 2]) - 1), int((a[:, 3] < a[:, 2]).sum()), ALT))
setup('1')
for q in n2.profile(1, N, 10, window=True):
    if q['layer'] in ('encoder/c1', 'encoder/c2'):
        say('#   %-11s %-44s grid=%-5d %7.2f us' % (q['layer'], q['name'], q['grid'][0], q['ms'] * 1e3))
Path(OUT).parent.mkdir(parents=True, exist_ok=True)
Path(OUT).write_text(''.join(lines))
