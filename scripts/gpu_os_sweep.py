#!/usr/bin/env python3
"""Round 5: the output-stationary kernel (ry_c2d_os) against the implicit GEMM + reduce on the six weight-streaming layers at the bottom
of the stage-2 U-Net -- per layer, every instantiated slice (tile rows / 4, tile channels / 4, waves, units in flight).

One process.  A slice is forced through RY_OS2 (re-read by ry_net_set_dtype, which also drops the launch plans); the layer's own launches are
timed with HIP events inside the eager window forward (ry_net_profile_window: the method behind profiles/*layers.txt), the whole stage-2
forward as graph replays of the convert call, and the chained two-lane step exactly as bench.py's step.  Results are checked against the
implicit-GEMM forward of the same window (other summation order only).

usage (GPU box): python scripts/gpu_os_sweep.py [frames] [out file] [reps]"""
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
OUT = sys.argv[2] if len(sys.argv) > 2 else str(ROOT / 'gpurun_out' / ('r5_os_sweep_n%d.txt' % N))
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 20
NAMES = ['encoder/c%d' % i for i in range(8)] + ['decoder/c%d' % i for i in range(8)]
LAYERS = tuple(int(v) for v in os.environ.get('SWEEP_LAYERS', '7,8,6,9,5,10').split(',') if v.strip().isdigit())      # 'none': only the planner's picks against the implicit GEMM
EMU = bool(os.environ.get('SWEEP_EMU'))                             # flow check on the CPU emulator (numbers mean nothing)

CONFIGS = [(m, n, w, d) for n in (1, 2, 4) for m in (1, 2, 3, 4, 6) for (w, d) in ((4, 4), (8, 4), (8, 2), (16, 2))]

(d1, P1), (d2, P2) = synth.model_params('SYN-8' if EMU else 'SYN-64')
if EMU:
    from realtime_yukarin_amd import _lib, build
    os.environ['RY_OS2_MINW'] = '1'
    ctx = engine.Context(0, _lib.Ry355Lib(build.build_emu()))
    LAYERS = (6,); CONFIGS = CONFIGS[:2]; REPS = 1
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


def setup(os2, maxm):
    os.environ['RY_OS2'] = os2
    os.environ['RY_OS2_MAXCOST'] = str(maxm)
    if not os2:
        del os.environ['RY_OS2']
    n2.set_dtype('f32')


def layer_us(reps=REPS):
    """microseconds per layer (all launches of the layer: GEMM + reduce), from events inside the eager window forward"""
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


say('# ry_c2d_os slice sweep, SYN-64, %d-frame window (T = %d); us per layer = all launches of the layer inside the eager window forward (HIP events, %d reps)' % (N, N + 128 - N % 128, REPS))
setup('', 0)
base = layer_us()
y0 = result()
f0 = forward_alone()
bot0 = sum(base[NAMES[l]][0] for l in (5, 6, 7, 8, 9, 10))
say('# implicit GEMM + reduce (RY_OS2_MAXCOST=0): stage-2 forward alone %.4f ms (graph replay); bottom six %.1f us' % (f0, bot0))
for l in (5, 6, 7, 8, 9, 10):
    say('#   %-11s %7.2f us   %s' % (NAMES[l], base[NAMES[l]][0], ' + '.join(base[NAMES[l]][1])))
best = {}
for layer in LAYERS:
    rows = []
    for c in CONFIGS:
        try:
            setup('%d:%d:%d:%d:%d' % ((layer,) + c), 0)
            lu = layer_us()
        except Exception as e:                                       # not a slice for this layer (K units, channels) or not instantiated
            continue
        names = lu[NAMES[layer]][1]
        if not names[0].startswith('ry_c2d_os<'):
            continue
        rows.append((lu[NAMES[layer]][0], c, names[0]))
    rows.sort()
    say('%s  (implicit GEMM + reduce %.2f us)' % (NAMES[layer], base[NAMES[layer]][0]))
    for us, c, nm in rows:
        say('    %-14s %7.2f us   %s' % ('%d:%d:%d:%d' % c, us, nm))
    if rows:
        best[layer] = rows[0]

# the winners together (only where they beat the implicit GEMM), checked against the implicit-GEMM result
use = {l: v for l, v in best.items() if v[0] < base[NAMES[l]][0]}
cfg = ','.join('%d:%d:%d:%d:%d' % ((l,) + v[1]) for l, v in sorted(use.items()))
say('# winners: RY_OS2=%s' % cfg)
setup(cfg, 0)
lu = layer_us()
y1 = result()
f1 = forward_alone()
err = float(numpy.abs(numpy.log(y1) - numpy.log(y0)).max() / numpy.abs(numpy.log(y0)).max())
say('# with the winners: stage-2 forward alone %.4f ms (was %.4f); bottom six %.1f us (was %.1f); log-spectrum rel diff to the implicit-GEMM forward %.3g; second run bit-identical: %s'
    % (f1, f0, sum(lu[NAMES[l]][0] for l in (5, 6, 7, 8, 9, 10)), bot0, err, numpy.array_equal(result(), y1)))
for l in (5, 6, 7, 8, 9, 10):
    say('#   %-11s %7.2f us   %s' % (NAMES[l], lu[NAMES[l]][0], ' + '.join(lu[NAMES[l]][1])))
# the default planner (RY_OS2 unset, RY_OS2_MAXCOST default)
os.environ.pop('RY_OS2', None); os.environ.pop('RY_OS2_MAXCOST', None); n2.set_dtype('f32')
lu = layer_us()
f2 = forward_alone()
say('# planner defaults: stage-2 forward alone %.4f ms; bottom six %.1f us' % (f2, sum(lu[NAMES[l]][0] for l in (5, 6, 7, 8, 9, 10))))
for l in (5, 6, 7, 8, 9, 10):
    say('#   %-11s %7.2f us   %s' % (NAMES[l], lu[NAMES[l]][0], ' + '.join(lu[NAMES[l]][1])))
# chained two-lane step: implicit GEMM / winners / planner defaults, interleaved twice
for rnd in range(1 if EMU else 2):
    setup('', 0); a = two_lane()
    setup(cfg, 0); b = two_lane()
    os.environ.pop('RY_OS2', None); os.environ.pop('RY_OS2_MAXCOST', None); n2.set_dtype('f32'); c = two_lane()
    say('# two-lane step, ms per window: implicit GEMM %.4f   winners %.4f   planner defaults %.4f' % (a, b, c))
Path(OUT).parent.mkdir(parents=True, exist_ok=True)
Path(OUT).write_text(''.join(lines))
