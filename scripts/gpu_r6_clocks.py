#!/usr/bin/env python3
"""Round 6: the shader clock the chip sustains under back-to-back stage-2 forwards and under the chained two-lane step, for the direct kernels, the Winograd
defaults and a forced plan set (tools/clock_spy.cpp: one wave samples s_memtime against the constant-rate counter every 50 us on its own stream).
usage (GPU box): python scripts/gpu_r6_clocks.py [frames] [out] [RY_WINO spec ...]"""
import ctypes
import os
import subprocess
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
OUT = sys.argv[2] if len(sys.argv) > 2 else str(ROOT / 'gpurun_out' / 'r6_clocks.txt')
SPECS = sys.argv[3:]
so = ROOT / 'tools' / 'libclock_spy.so'
if not so.exists():
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', str(ROOT / 'tools' / 'clock_spy.cpp'), '-o', str(so)], check=True)
spy = ctypes.CDLL(str(so))
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
lines = []


def say(s):
    lines.append(s + '\n'); print(s, flush=True)
    open(OUT, 'w').writelines(lines)


def spied(run, seconds=0.25):
    """run(seconds) under the spy -> (ms per unit as run reports it, MHz percentiles over the busy stretch)"""
    khz = spy.spy_start(50, int((seconds + 0.1) * 1e6))
    assert khz > 0, khz
    time.sleep(0.02)
    ms = run(seconds)
    buf = (ctypes.c_ulonglong * (2 * 20000))()
    n = spy.spy_collect(buf, 20000)
    a = numpy.frombuffer(buf, dtype=numpy.uint64, count=2 * n).reshape(n, 2).astype(numpy.float64)
    dw, dc = numpy.diff(a[:, 0]), numpy.diff(a[:, 1])
    mhz = dc / numpy.maximum(dw, 1) * (khz / 1e3)
    lo, hi = int(0.03 / 50e-6), int((0.02 + seconds * 0.9) / 50e-6)       # the stretch in which the workload certainly runs
    m = mhz[lo:hi]
    return ms, (float(numpy.percentile(m, 5)), float(numpy.median(m)), float(numpy.percentile(m, 95)), float(mhz[:200].mean()))


def forwards(seconds):
    for _ in range(3):
        n2.convert_device(d_in, d_out, 1, N)
    ctx.sync()
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            n2.convert_device(d_in, d_out, 1, N)
        ctx.sync(); k += 20
    return (time.perf_counter() - t0) / k * 1e3


def two_lane_factory():
    core = engine.VcCore(n1, n2, mtx, lanes=2)
    k = [0]

    def run(seconds):
        def step():
            core.enqueue_device(d_x, d_rows, N, N, d_mc[k[0] % 6], d_sp[k[0] % 6], 1e-16); k[0] += 1
        for _ in range(18):
            step()
        ctx.sync()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(40):
                step()
            ctx.sync(); n += 40
        return (time.perf_counter() - t0) / n * 1e3
    return core, run


say('# sustained shader clock (MHz: 5th percentile / median / 95th percentile over the busy stretch; idle before it) next to the time per forward / per window, SYN-64, %d frames' % N)
for tag, wino, spec in [('direct', '0', '')] + [('winograd defaults', '1', '')] + [('RY_WINO=' + s, '1', s) for s in SPECS] + [('direct again', '0', '')]:
    os.environ['RY_WINOGRAD'] = wino
    if spec:
        os.environ['RY_WINO'] = spec
    else:
        os.environ.pop('RY_WINO', None)
    ctx.reload_env(); n2.set_dtype('f32')
    ms, c = spied(forwards)
    say('%-60s forward alone %.4f ms   clock %4.0f / %4.0f / %4.0f MHz (idle %4.0f)' % (tag, ms, c[0], c[1], c[2], c[3]))
    core, run = two_lane_factory()
    ms, c = spied(run)
    core.close()
    say('%-60s two-lane step %.4f ms   clock %4.0f / %4.0f / %4.0f MHz (idle %4.0f)' % (tag, ms, c[0], c[1], c[2], c[3]))
