import sys, time
from pathlib import Path
import numpy
sys.path.insert(0, '.')
import torch
torch.cuda.set_device(0)
from realtime_yukarin_amd import engine, sptk, synth
from realtime_yukarin_amd.weights import synthetic_params, flatten_params
N = 300
ctx = engine.get_context(0)
d1, d2 = synth.model_descs('SYN-64')
n1 = engine.Net(ctx, d1, flatten_params(d1, synthetic_params(d1, synth.SEED_STAGE1)))
n2 = engine.Net(ctx, d2, flatten_params(d2, synthetic_params(d2, synth.SEED_STAGE2)), width=synth.FFT_BINS - 1)
mtx = sptk.mc2sp_matrix(d1.out_ch - 1, sptk.mcepalpha(16000), 2 * (synth.FFT_BINS - 1))
x = synth.stage1_input(N, 1)[0]
d_x = ctx.dev_alloc(N * d1.in_ch); ctx.dev_upload(d_x, x)
d_rows = ctx.dev_alloc(N); ctx.dev_upload(d_rows, numpy.arange(N, dtype=numpy.int32))
outs = [(ctx.dev_alloc(N * d1.out_ch), ctx.dev_alloc(N * synth.FFT_BINS)) for _ in range(6)]
for lanes in (1, 2):
    core = engine.VcCore(n1, n2, mtx, lanes=lanes)
    for i in range(24):
        core.enqueue_device(d_x, d_rows, N, N, outs[i % 6][0], outs[i % 6][1], 1e-16)
    ctx.sync()
    ts = []
    for i in range(24):
        t0 = time.perf_counter()
        core.enqueue_device(d_x, d_rows, N, N, outs[i % 6][0], outs[i % 6][1], 1e-16)
        ctx.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    print('lanes', lanes, 'rotating outputs, sync each:', ' '.join('%.2f' % t for t in ts))
    ts = []
    for i in range(24):
        t0 = time.perf_counter()
        core.enqueue_device(d_x, d_rows, N, N, outs[0][0], outs[0][1], 1e-16)
        ctx.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    print('lanes', lanes, 'fixed outputs, sync each:   ', ' '.join('%.2f' % t for t in ts))
    core.close()
