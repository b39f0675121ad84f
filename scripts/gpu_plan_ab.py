#!/usr/bin/env python3
"""Stage-2 launch plans under two window lanes, in turn (env assignments separated by ";" so that RY_WINO lists fit).  The planner prices a layer by its lone time (one workgroup of two K groups per CU: 8 waves,
124 KiB of LDS); two lanes put two launches on the chip, and two such workgroups do not fit one CU.  Candidates (RY_PLAN strings, "-" = the planner's
picks) are measured in turn: stage-2 forward alone (graph replay) and the chained two-lane step exactly as bench.py's step.
usage (GPU box): python scripts/gpu_plan_ab.py [frames] [alternations] [out file] plan [plan ...]"""
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
OUT = sys.argv[3] if len(sys.argv) > 3 else str(ROOT / 'gpurun_out' / ('r6_plan_ab_n%d.txt' % N))
PLANS = sys.argv[4:] or ['-']
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


ENV_KEYS = set()


def setup(plan):
    """'-': the planner's picks; 'env:K=V[,K=V]': the planner's picks under these switches; else an RY_PLAN string"""
    for k in ENV_KEYS:
        os.environ.pop(k, None)
    os.environ.pop('RY_PLAN', None)
    if plan.startswith('env:'):
        for kv in plan[4:].split(';'):
            k, v = kv.split('=', 1)
            os.environ[k] = v; ENV_KEYS.add(k)
    elif plan != '-':
        os.environ['RY_PLAN'] = plan
    reread(); n2.set_dtype('f32')


def forward_alone(reps=40):
    for _ in range(3):
        n2.convert_device(d_in, d_out, 1, N)
    ctx.sync(); ctx.timer_start()
    for _ in range(reps):
        n2.convert_device(d_in, d_out, 1, N)
    return ctx.timer_stop() / reps


def two_lane(steps=100):
    core = engine.VcCore(n1, n2, mtx, lanes=int(os.environ.get('AB_LANES', '2')))
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


say('# stage-2 plans under two lanes, SYN-64, %d frames, %d rounds over %d candidates' % (N, ALT, len(PLANS)))
res = {p: [] for p in PLANS}
for r in range(ALT):
    for p in PLANS:
        setup(p); f = forward_alone(); t = two_lane()
        res[p].append((f, t))
        say('%2d  %-60s forward alone %.4f ms   two-lane step %.4f ms per window' % (r, p, f, t))
for p in PLANS:
    a = numpy.array(res[p])
    say('# median  %-60s forward alone %.4f ms   two-lane step %.4f ms' % (p, numpy.median(a[:, 0]), numpy.median(a[:, 1])))
    setup(p)
    for q in n2.profile(1, N, 5, window=True):
        if q['name'].startswith(('ry_igemm', 'ry_wino', 'ry_splitk', 'ry_rep')) and (q['ms'] > 0.02 or q['name'].startswith('ry_wino')):
            say('#     %-11s %-44s grid=%-5d %7.2f us' % (q['layer'], q['name'], q['grid'][0], q['ms'] * 1e3))
Path(OUT).parent.mkdir(parents=True, exist_ok=True)
Path(OUT).write_text(''.join(lines))
