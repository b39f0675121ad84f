"""Soak run (GPU box): thousands of windows of changing length, silence pattern and batch through the drop-in entry points
(`ry_vc_convert`, `ry_sr_convert`), exercising plan-cache eviction and hipGraph re-capture; a fixed probe window must come
back bit-identical every time and the free device memory must not trend down (it saw-tooths with the bounded plan cache).  Usage: python scripts/gpu_soak.py [iterations]"""
import sys
import time
from pathlib import Path

import numpy

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch                                                                     # noqa: E402
from realtime_yukarin_amd import engine, sptk, synth                             # noqa: E402
from realtime_yukarin_amd.weights import flatten_params                          # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
(d1, P1), (d2, P2) = synth.model_params('SYN-64')
ctx = engine.get_context(0)
n1 = engine.Net(ctx, d1, flatten_params(d1, P1))
n2 = engine.Net(ctx, d2, flatten_params(d2, P2), width=synth.FFT_BINS - 1)
core = engine.VcCore(n1, n2, sptk.mc2sp_matrix(8, sptk.mcepalpha(16000), 1024))
rng = numpy.random.default_rng(7)
probe_x = synth.stage1_input(300)[0]
probe_eff = rng.random(300) > 0.2
probe = core.convert(probe_x[probe_eff], probe_eff)
probe_sp = synth.stage2_input(200, windows=3)
probe_b = n2.convert(probe_sp)
free0 = None
samples = []
t0 = time.time()
frames_done = 0
for it in range(iters):
    n = int(rng.integers(40, 700))
    kind = it % 7
    if kind == 5:                                       # the silence gate on the device: raw wave + all frames up, mask back (ry_vc_submit_wave)
        from realtime_yukarin_amd import gate
        w = (rng.choice([1e-5, 0.1]) * rng.normal(size=n * 80)).astype('f4'); w[(n // 3) * 80:(n // 2) * 80] = 0.0
        feat = rng.normal(size=(n, d1.in_ch)).astype('f4')
        mc, sp, eff = core.wait_wave(core.submit_wave(w, 80, 1024, *gate.thresholds(60), feat))
        assert sp.shape == (n, synth.FFT_BINS) and numpy.isfinite(sp).all() and not mc[~eff].any() and not eff[n // 3 + 7:n // 2 - 7].any()
    elif kind == 6:                                     # a short stream with three windows in flight through the pinned ring
        wins = []
        for _ in range(4):
            m = int(rng.integers(40, 400)); e = rng.random(m) < 0.8
            wins.append((rng.normal(size=(m, d1.in_ch)).astype('f4')[e], e))
        for (xe, e), (mc, sp) in zip(wins, core.convert_stream(wins, depth=3)):
            assert sp.shape == (len(e), synth.FFT_BINS) and numpy.isfinite(sp).all() and not mc[~e].any()
    elif kind < 3:                                      # one window through the fused core, data-dependent effective count
        eff = rng.random(n) < rng.choice([0.0, 0.5, 0.9, 1.0], p=[0.05, 0.25, 0.4, 0.3])
        x = rng.normal(size=(n, d1.in_ch)).astype('f4')
        if kind == 2:                                   # ... the caller discards frames at both ends (ry_vc_set_discard): kept rows equal the full call
            front, back = int(rng.integers(0, n // 2)), int(rng.integers(0, n // 3))
            full = core.convert(x[eff], eff)
            core.set_discard(front, back)
            mc, sp = core.convert(x[eff], eff)
            core.set_discard(0, 0)
            k1 = n - back if n - back > front else n
            k0 = front if k1 > front else 0
            assert numpy.array_equal(sp[k0:k1], full[1][k0:k1]) and not sp[:k0].any() and not sp[k1:].any() and numpy.array_equal(mc, full[0]), (n, front, back)
            sp = full[1]
        else:
            mc, sp = core.convert(x[eff], eff)
        assert sp.shape == (n, synth.FFT_BINS) and numpy.isfinite(sp).all() and not mc[~eff].any()
    else:                                               # a batch of windows through stage 2 alone (plan cache keys: batch x length)
        b = int(rng.integers(1, 5))
        sp = n2.convert(synth.stage2_input(min(n, 400), windows=b, seed=it))
        assert sp.shape == (b, min(n, 400), synth.FFT_BINS) and numpy.isfinite(sp).all()
    frames_done += n
    if it % 100 == 99 or it == iters - 1:
        again = core.convert(probe_x[probe_eff], probe_eff)
        assert numpy.array_equal(again[0], probe[0]) and numpy.array_equal(again[1], probe[1]), 'probe window changed at %d' % it
        assert numpy.array_equal(n2.convert(probe_sp), probe_b), 'probe batch changed at %d' % it
        free = torch.cuda.mem_get_info()[0]
        free0 = free0 or free
        samples.append(free)
        print('iter %5d  %.1f s  %.0f windows/s  free HBM %.2f GB (drift %+.1f MB)' % (
            it + 1, time.time() - t0, (it + 1) / (time.time() - t0), free / 1e9, (free - free0) / 1e6), flush=True)
# the plan cache (<= 16 plans per predictor of up to ~0.5 GB each, cleared when full) makes the free memory saw-tooth by several GB;
# it must not TREND down: compare the low-water marks of the last and the first third of the run
third = max(1, len(samples) // 3)
assert min(samples[-third:]) >= min(samples[:third]) - 2e9, 'device memory keeps shrinking: %s' % [round(v / 1e9, 2) for v in samples]
print('soak ok: %d windows, %d frames, probes bit-identical throughout' % (iters, frames_done))
core.close(); n1.close(); n2.close()
