"""Experiment: K independent window cores (each its own pair of predictor streams and activation buffers) fed round robin --
does a second / third stage-2 stream fill the tails of the one-round grids and the bottom layers of the other's forward?
usage: python scripts/gpu_r2_twostream.py [cores[:lanes] ...]   (lanes = ry_vc_set_lanes of every core: clones that share the filters)"""
import sys, time
from pathlib import Path
import numpy
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os
import torch  # noqa: F401  (one HIP runtime)
if os.environ.get('TS_TORCH_INIT') == '1':
    torch.cuda.set_device(0); torch.cuda.synchronize(); _t = torch.zeros(8, device='cuda')
from realtime_yukarin_amd import engine, sptk, synth
from realtime_yukarin_amd.weights import synthetic_params, flatten_params

N = 300
ctx = engine.get_context(0)
d1, d2 = synth.model_descs('SYN-64')
b1 = flatten_params(d1, synthetic_params(d1, synth.SEED_STAGE1)); b2 = flatten_params(d2, synthetic_params(d2, synth.SEED_STAGE2))
mtx = sptk.mc2sp_matrix(d1.out_ch - 1, sptk.mcepalpha(16000), 2 * (synth.FFT_BINS - 1))
x = synth.stage1_input(N, 1)[0]
d_x = ctx.dev_alloc(N * d1.in_ch); ctx.dev_upload(d_x, x)
d_rows = ctx.dev_alloc(N); ctx.dev_upload(d_rows, numpy.arange(N, dtype=numpy.int32))
for K, L in [tuple(int(v) for v in (a + ':1').split(':')[:2]) for a in sys.argv[1:]] or [(1, 1), (2, 1), (3, 1)]:
    cores, nets, outs = [], [], []
    for k in range(K):
        n1 = engine.Net(ctx, d1, b1); n2 = engine.Net(ctx, d2, b2, width=synth.FFT_BINS - 1)
        nets += [n1, n2]; cores.append(engine.VcCore(n1, n2, mtx, lanes=L))
        outs.append([(ctx.dev_alloc(N * d1.out_ch), ctx.dev_alloc(N * synth.FFT_BINS)) for _ in range(6)])
    def run(steps):
        for i in range(steps):
            c = cores[i % K]; mc, sp = outs[i % K][(i // K) % 6]
            c.enqueue_device(d_x, d_rows, N, N, mc, sp, 1e-16)
    res = []
    for rep in range(3):
        run(12); ctx.sync()
        t0 = time.perf_counter(); run(120); ctx.sync(); res.append((time.perf_counter() - t0) / 120 * 1e3)
    sp0 = numpy.empty((N, synth.FFT_BINS), numpy.float32); ctx.dev_download(outs[0][0][1], sp0)
    spk = numpy.empty((N, synth.FFT_BINS), numpy.float32); ctx.dev_download(outs[K - 1][0][1], spk)
    print('cores %d x lanes %d: ms per window %s  (same result on every core: %s)' % (K, L, ' '.join('%.4f' % r for r in res), numpy.array_equal(sp0, spk)), flush=True)
    for c in cores: c.close()
    for n in nets: n.close()
