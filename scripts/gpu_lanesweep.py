#!/usr/bin/env python3
"""Round 3: launch plans of the eight MFMA-bound stage-2 layers chosen UNDER TWO WINDOW LANES.

The planner (choose_igemm) and the round-1 / round-2 sweeps price a layer running ALONE on the chip.  The headline step runs two windows
side by side (ry_vc_set_lanes): the kernels of two forwards share the CUs, a one-round grid of 512-thread workgroups leaves no room for
the other lane's workgroups, tails and epilogues of one lane are (or are not) covered by the other.  This script measures what the
step itself says: one process, coordinate descent over the layers, every (tile, K groups, external splits) candidate forced through
RY_PLAN (re-read by ry_net_set_dtype), the window core re-created (fresh clones, plans, graphs), then the chained two-lane step timed
exactly like bench.py does (priming of every ring slot, warm-up, K steps between synchronisations).  Also prints the single-window
forward (graph replay of stage 2 alone) of every candidate, because a plan that wins with two lanes may lose alone.

usage (GPU box): python scripts/gpu_lanesweep.py [frames] [out file] [steps]"""
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
OUT = sys.argv[2] if len(sys.argv) > 2 else str(ROOT / 'gpurun_out' / ('r3_lanesweep_n%d.txt' % N))
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 80
NAMES = ['encoder/c%d' % i for i in range(8)] + ['decoder/c%d' % i for i in range(8)]
TILES = {1: '128x128', 6: '96x128', 3: '64x128', 5: '128x64', 4: '32x128'}
LAYERS = tuple(int(v) for v in os.environ.get('SWEEP_LAYERS', '12,13,14,11,1,2,3,4').split(','))     # default: the eight MFMA-bound layers, biggest first
SWEEP_TILES = tuple(int(v) for v in os.environ['SWEEP_TILES'].split(',')) if os.environ.get('SWEEP_TILES') else None
SWEEP_SPLITS = tuple(int(v) for v in os.environ.get('SWEEP_SPLITS', '1,2,3,4').split(','))

EMU = bool(os.environ.get('SWEEP_EMU'))                             # flow check of this script on the CPU emulator (numbers mean nothing)
(d1, P1), (d2, P2) = synth.model_params('SYN-8' if EMU else 'SYN-64')
if EMU:
    from realtime_yukarin_amd import _lib, build
    ctx = engine.Context(0, _lib.Ry355Lib(build.build_emu()))
    LAYERS = (12,)
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
ctx.dev_upload(d_in, synth.stage2_input(N)[0])


def measure(plan, steps=STEPS, lanes=2):
    """(ms per window of the chained step with `lanes` lanes, ms of one stage-2 forward alone) under RY_PLAN=plan."""
    os.environ['RY_PLAN'] = plan
    n2.set_dtype('f32')                                              # re-reads RY_PLAN, drops the plans and graphs
    core = engine.VcCore(n1, n2, mtx, lanes=lanes)
    k = [0]

    def step():
        core.enqueue_device(d_x, d_rows, N, N, d_mc[k[0] % 6], d_sp[k[0] % 6], 1e-16)
        k[0] += 1
    for _ in range(2 if EMU else 12 + 6):
        step()
    ctx.sync()
    best = 1e9
    for _ in range(2):
        for _ in range(4):
            step()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    for _ in range(3):
        n2.convert_device(d_in, d_out, 1, N)
    ctx.sync(); ctx.timer_start()
    for _ in range(20):
        n2.convert_device(d_in, d_out, 1, N)
    alone = ctx.timer_stop() / 20
    core.close()
    return best, alone


lines = []


def say(s):
    lines.append(s + '\n')
    print(s, flush=True)


base = [measure('') for _ in range(3)]
b2 = min(b[0] for b in base); b1 = min(b[1] for b in base)
say('# two-lane plan sweep, SYN-64, %d frames; ms per window of the chained two-lane step / ms of one stage-2 forward alone' % N)
say('# planner: %s' % '  '.join('%.4f/%.4f' % b for b in base))
chosen = {}


def plan_str(extra=None):
    d = dict(chosen)
    if extra:
        d[extra[0]] = extra[1]
    return ','.join('%d:%s' % (l, v) for l, v in sorted(d.items()))


cur = b2
for layer in LAYERS:
    cands = []
    tiles = SWEEP_TILES if SWEEP_TILES else ((5,) if layer == 14 else (6, 1, 3))
    for tile in tiles[:1] if EMU else tiles:
        for kg in ((1,) if EMU else (1, 2)):
            for sp in (1,) if EMU else SWEEP_SPLITS:
                cands.append('%d:%d:%d' % (tile, sp, kg))
    best = (cur, None)
    for c in cands:
        try:
            t2, t1 = measure(plan_str((layer, c)))
        except Exception as e:
            say('%-11s %-20s refused: %s' % (NAMES[layer], c, str(e)[:70]))
            continue
        say('%-11s %-8s s%s kg%s   %.4f  %.4f' % (NAMES[layer], TILES[int(c.split(':')[0])], c.split(':')[1], c.split(':')[2], t2, t1))
        if t2 < best[0]:
            best = (t2, c)
    if best[1] is not None and best[0] < cur * 0.994:                # at least 0.6 %: above the run-to-run noise of the step
        t2, t1 = measure(plan_str((layer, best[1])))                 # confirm
        if t2 < cur * 0.996:
            chosen[layer] = best[1]; cur = 0.5 * (t2 + best[0])
            say('%-11s TAKE %s -> %.4f (alone %.4f)' % (NAMES[layer], best[1], t2, t1))
            continue
    say('%-11s keep the planner pick (%.4f)' % (NAMES[layer], cur))
final = plan_str()
say('# RY_PLAN=%s' % final)
for i in range(3):
    a = measure(''); b = measure(final)
    say('# interleaved %d: planner %.4f / %.4f   swept %.4f / %.4f' % (i, a[0], a[1], b[0], b[1]))
Path(OUT).parent.mkdir(parents=True, exist_ok=True)
open(OUT, 'w').writelines(lines)
n1.close(); n2.close()
