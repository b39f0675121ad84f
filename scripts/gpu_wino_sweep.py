#!/usr/bin/env python3
"""Round 6: the Winograd F(2x2, 2x2) kernels (ry_wino_ldsdma) against the direct implicit GEMM on the eight MFMA-bound stage-2 layers.

One process.  A plan is forced through RY_WINO="layer:cfg:mbw:splits" (re-read by ry_net_set_dtype, which also drops the launch plans); the layer's own
launches are timed with HIP events inside the eager window forward (ry_net_profile_window), the whole stage-2 forward as graph replays of the convert
call, and the chained two-lane step exactly as bench.py's step.  Results are checked against the direct forward of the same window.

usage (GPU box): python scripts/gpu_wino_sweep.py [frames] [out file] [reps]      (SWEEP_LAYERS=none: only the planner's defaults)"""
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
OUT = sys.argv[2] if len(sys.argv) > 2 else str(ROOT / 'gpurun_out' / ('r6_wino_n%d.txt' % N))
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 10
NAMES = ['encoder/c%d' % i for i in range(8)] + ['decoder/c%d' % i for i in range(8)]
LAYERS = tuple(int(v) for v in os.environ.get('SWEEP_LAYERS', '12,13,14,11,1,2,3,4').split(',') if v.strip().isdigit())
BY_FORWARD = os.environ.get('SWEEP_BY', 'forward') in ('forward', 'lanes')       # rank a layer's plans by the graph-replayed forward they give (default) or by the layer's own eager launches
BY_LANES = os.environ.get('SWEEP_BY', 'forward') == 'lanes'           # ... by the chained two-lane step (bench.py's step) with every OTHER layer on the planner's default: what the headline runs
EMU = bool(os.environ.get('SWEEP_EMU'))                             # flow check on the CPU emulator (numbers mean nothing)
SPLITS = tuple(int(v) for v in os.environ.get('SWEEP_SPLITS', '0,1,2,3,4,5,6,8').split(','))
CFGS = tuple(int(v) for v in os.environ.get('SWEEP_CFGS', '1,2,3,4').split(','))     # 1: 2 x 2 waves, 2: 4 x 2 waves (two slices per barrier), 3 / 4: two channel blocks per wave, one wave per SIMD (2 x 2 / 4 x 1 waves)
CONFIGS = [(c, m, s) for c in CFGS for m in ((1, 2) if c in (1, 3) else (1, 2, 4)) for s in SPLITS]

(d1, P1), (d2, P2) = synth.model_params('SYN-8' if EMU else 'SYN-64')
if EMU:
    from realtime_yukarin_amd import _lib, build
    os.environ['RY_WINO_MINM'] = '1'
    ctx = engine.Context(0, _lib.Ry355Lib(build.build_emu()))
    LAYERS = LAYERS[:1]; CONFIGS = CONFIGS[:2]; REPS = 1
else:
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
sp_in = synth.stage2_input(N)[0]
ctx.dev_upload(d_in, sp_in)
lines = []


def say(s):
    lines.append(s + '\n')
    print(s, flush=True)
    with open(OUT, 'w') as f:
        f.writelines(lines)


def setup(wino, on=1):
    os.environ['RY_WINOGRAD'] = str(on)
    if wino:
        os.environ['RY_WINO'] = wino
    else:
        os.environ.pop('RY_WINO', None)
    n2.set_dtype('f32')


def layer_us(reps=REPS):
    """microseconds per layer (all launches of the layer: GEMM + reduce + copied rows), from events inside the eager window forward"""
    if not EMU:
        n2.profile(1, N, 2, window=True)
    st = n2.profile(1, N, reps, window=True)
    out = {}
    for q in st:
        out.setdefault(q['layer'], [0.0, []])
        out[q['layer']][0] += q['ms'] * 1e3
        out[q['layer']][1].append('%s grid=%d' % (q['name'], q['grid'][0]))
    return out


def forward_alone(reps=30):
    reps = 1 if EMU else reps
    for _ in range(1 if EMU else 3):
        n2.convert_device(d_in, d_out, 1, N)
    ctx.sync(); ctx.timer_start()
    for _ in range(reps):
        n2.convert_device(d_in, d_out, 1, N)
    return ctx.timer_stop() / reps


def result():
    n2.convert_device(d_in, d_out, 1, N)
    ctx.sync()
    y = numpy.empty((N, 513), numpy.float32)
    ctx.dev_download(d_out, y)
    return y


def two_lane(steps=80):
    steps = 1 if EMU else steps
    core = engine.VcCore(n1, n2, mtx, lanes=2)
    k = [0]

    def step():
        core.enqueue_device(d_x, d_rows, N, N, d_mc[k[0] % 6], d_sp[k[0] % 6], 1e-16)
        k[0] += 1
    for _ in range(2 if EMU else 18):
        step()
    ctx.sync()
    best = 1e9
    for _ in range(1 if EMU else 3):
        for _ in range(1 if EMU else 4):
            step()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    core.close()
    return best


say('# ry_wino_ldsdma plan sweep, SYN-64, %d-frame window (T = %d); us per layer = all launches of the layer inside the eager window forward (HIP events, %d reps)' % (N, N + 128 - N % 128, REPS))
setup('', 0)
base = layer_us()
y0 = result()
f0 = forward_alone()
if BY_LANES:
    setup('', 1)
    f0 = min(two_lane(40) for _ in range(3))          # lanes: the reference is the two-lane step under the planner's own picks
say('# direct implicit GEMM (RY_WINOGRAD=0): stage-2 forward alone %.4f ms (graph replay)' % f0)
for l in (1, 2, 3, 4, 11, 12, 13, 14):
    say('#   %-11s %7.2f us   %s' % (NAMES[l], base[NAMES[l]][0], ' + '.join(base[NAMES[l]][1])))
best = {}
for layer in LAYERS:
    rows = []
    for c in CONFIGS:
        try:
            setup(','.join(([] if BY_LANES else ['%d:0' % l for l in (1, 2, 3, 4, 11, 12, 13, 14) if l != layer]) + ['%d:%d:%d:%d' % ((layer,) + c)]))
            lu = layer_us()
        except Exception as e:                                       # no such plan for this layer
            if 'no Winograd plan' not in str(e) and 'RY_WINO' not in str(e):
                say('%-11s cfg %d mbw %d splits %d: %s' % (NAMES[layer], c[0], c[1], c[2], str(e)[:120]))
            continue
        us, names = lu[NAMES[layer]]
        fwd = (two_lane(40) if BY_LANES else forward_alone(20)) * 1e3     # the whole stage-2 forward under graph replay with this one layer in Winograd form (lanes: the two-lane step, the other layers on their defaults)
        rows.append((fwd if BY_FORWARD else us, c, names))
        say('%-11s cfg %d mbw %d splits %2d  %7.2f us  (direct %7.2f)  forward %8.2f us (direct %8.2f)   %s' % (NAMES[layer], c[0], c[1], c[2], us, base[NAMES[layer]][0], fwd, f0 * 1e3, ' + '.join(names)))
    if rows:
        rows.sort()
        best[layer] = rows[0]
        say('# best %-11s cfg %d mbw %d splits %d: %.2f us against %.2f direct' % ((NAMES[layer],) + rows[0][1] + (rows[0][0], f0 * 1e3 if BY_FORWARD else base[NAMES[layer]][0])))
if best:
    spec = ','.join('%d:%d:%d:%d' % ((l,) + best[l][1]) if best[l][0] < (f0 * 1e3 * (0.993 if BY_LANES else 1.0) if BY_FORWARD else base[NAMES[l]][0]) else ('' if BY_LANES else '%d:0' % l) for l in sorted(best))
    spec = ','.join(v for v in spec.split(',') if v)
    setup(spec)
    yb = result()
    fb = forward_alone()
    say('# the per-layer winners together (RY_WINO=%s): forward alone %.4f ms; result vs direct: max |y / y0 - 1| = %.3g' % (spec, fb, float(numpy.abs(yb / y0 - 1).max())))
    say('#   two-lane step %.4f ms per window' % two_lane())
setup('', 1)
lu = layer_us()
yd = result()
fd = forward_alone()
say('# the planner\'s defaults (RY_WINOGRAD=1): forward alone %.4f ms; result vs direct: max |y / y0 - 1| = %.3g' % (fd, float(numpy.abs(yd / y0 - 1).max())))
for l in (1, 2, 3, 4, 11, 12, 13, 14):
    say('#   %-11s %7.2f us   %s' % (NAMES[l], lu[NAMES[l]][0], ' + '.join(lu[NAMES[l]][1])))
t1 = two_lane()
setup('', 0)
t0 = two_lane()
say('# two-lane step: Winograd defaults %.4f ms per window, direct %.4f ms per window' % (t1, t0))
