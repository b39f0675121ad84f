"""Parity under weights with TRAINED-model statistics (oracle/trained_like.py): BatchNormalization `avg_var` spanning decades (the
folded per-channel scale gamma / sqrt(avg_var + eps) from ~ 1e-2 to > 1e2), gamma of both signs, non-zero conv bias / beta / avg_mean.
Every other full-size test uses the pix2pix-init synthetic weights, whose folded scales are all ~ 1.  Loaded by the constructors the
reference calls at /root/reference/realtime_voice_conversion/converter/yukarin_converter.py:40-55.  Emulator (SYN-8, CPU suite) and
the real GPU (SYN-64, -m gpu), both wrappers, against the torch / oneDNN oracle AND its float64 evaluation (so that the bar is not
spent on the oracle's own fp32 rounding)."""
import numpy
import pytest
import torch

from oracle import mc2sp as omc
from oracle import torch_ref, trained_like
from realtime_yukarin_amd import engine, sptk, synth
from realtime_yukarin_amd.weights import flatten_params, synthetic_params, validate_params

TOL = 1e-4


def calib_inputs(n_frames):
    x = synth.stage1_input(n_frames, seed=811)[0]
    x1 = numpy.pad(x.T, [(0, 0), (0, 128 - n_frames % 128)], mode='minimum')[numpy.newaxis]
    sp = synth.stage2_input(n_frames, seed=812)[0]
    x2 = numpy.log(numpy.pad(sp, [(0, 128 - n_frames % 128), (0, 0)], mode='minimum'))[:, :-1][numpy.newaxis, numpy.newaxis]
    return x1, x2


def run(ctx, name, n_frames, windows):
    d1, d2 = synth.model_descs(name)
    x1c, x2c = calib_inputs(n_frames)
    P1 = trained_like.calibrated_params(d1, 821, x1c)
    P2 = trained_like.calibrated_params(d2, 822, x2c)
    validate_params(d1, P1); validate_params(d2, P2)
    for P in (P1, P2):
        st = trained_like.describe(P)
        print(name, st)
        assert st['avg_var'][1] / st['avg_var'][0] > 1e3 and st['folded_scale'][1] / st['folded_scale'][0] > 1e2 and st['abs_gamma'][1] > 1.5
    n1 = engine.Net(ctx, d1, flatten_params(d1, P1))
    n2 = engine.Net(ctx, d2, flatten_params(d2, P2), width=synth.FFT_BINS - 1)
    t1, t2 = torch_ref.TorchUNet(P1), torch_ref.TorchUNet(P2)
    t1d, t2d = torch_ref.TorchUNet(P1, dtype=torch.float64), torch_ref.TorchUNet(P2, dtype=torch.float64)
    mtx = sptk.mc2sp_matrix(d1.out_ch - 1, sptk.mcepalpha(16000), 1024)
    P1s = synthetic_params(d1, synth.SEED_STAGE1)
    n1s = engine.Net(ctx, d1, flatten_params(d1, P1s))
    t1sd = torch_ref.TorchUNet(P1s, dtype=torch.float64)
    try:
        for n in windows:                                             # other windows than the calibration one
            x = synth.stage1_input(n, seed=830 + n)[0]
            y = n1.convert(x)
            r, rd = torch_ref.stage1_convert_core(t1, x), torch_ref.stage1_convert_core(t1d, x.astype(numpy.float64))
            e32, e64 = (float(numpy.abs(y - q).max() / numpy.abs(q).max()) for q in (r, rd))
            o32 = float(numpy.abs(r - rd).max() / numpy.abs(rd).max())
            print('%s stage-1 n=%d: vs fp32 oracle %.2e, vs float64 oracle %.2e (fp32 oracle vs float64: %.2e)' % (name, n, e32, e64, o32))
            assert e32 < TOL and e64 < TOL and float(numpy.abs(rd).max()) > 1e-2
            sp = synth.stage2_input(n, seed=840 + n)[0]
            y = n2.convert(sp).astype(numpy.float64)
            r, rd = torch_ref.stage2_convert(t2, sp), torch_ref.stage2_convert(t2d, sp.astype(numpy.float64))
            e32, e64 = float(numpy.abs(y / r - 1).max()), float(numpy.abs(y / rd - 1).max())
            print('%s stage-2 n=%d: element-wise vs fp32 oracle %.2e, vs float64 oracle %.2e (fp32 oracle vs float64: %.2e); log-spectrum range %.2f .. %.2f'
                  % (name, n, e32, e64, float(numpy.abs(r / rd - 1).max()), float(numpy.log(rd).min()), float(numpy.log(rd).max())))
            assert e32 < TOL and e64 < TOL and numpy.isfinite(y).all()
            assert float(numpy.log(rd).max() - numpy.log(rd).min()) > 0.05             # the predictor did something
            # round 6: the default stage-2 forward runs its MFMA-bound layers in Winograd F(2x2, 2x2) form; the direct kernels (RY_WINOGRAD=0) on THESE weights
            # are the reference the verdict holds it to: 1e-5 on the log-spectrum (the quantity the predictor computes), and the direct form meets the oracle too
            import os
            os.environ['RY_WINOGRAD'] = '0'; ctx.reload_env(); n2.set_dtype('f32')
            try:
                ydir = n2.convert(sp).astype(numpy.float64)
            finally:
                del os.environ['RY_WINOGRAD']; ctx.reload_env(); n2.set_dtype('f32')
            ew = float(numpy.abs(numpy.log(y) - numpy.log(ydir)).max() / numpy.abs(numpy.log(ydir)).max())
            print('%s stage-2 n=%d: Winograd form vs direct form: log-spectrum %.2e (relative to its maximum), element-wise %.2e; direct vs float64 oracle %.2e; identical: %s'
                  % (name, n, ew, float(numpy.abs(y / ydir - 1).max()), float(numpy.abs(ydir / rd - 1).max()), numpy.array_equal(y, ydir)))
            assert ew < 1e-5 and float(numpy.abs(ydir / rd - 1).max()) < TOL
            # round 5: the split-bf16 mode ('bf16x3': x w ~ x_hi w_hi + x_lo w_hi + x_hi w_lo) on THESE weights -- folded scales from 3e-4 to
            # 4e2 are where a dropped lo * lo term or a mis-scaled split would show -- held to the same 1e-4 bar, stage 2 alone and through
            # the chained window core (stage 1 -> combine_silent -> mc2sp -> stage 2)
            n2.set_dtype('bf16x3')
            y3 = n2.convert(sp).astype(numpy.float64)
            e3 = float(numpy.abs(y3 / rd - 1).max())
            print('%s stage-2 n=%d in split-bf16 mode: element-wise vs float64 oracle %.2e (differs from the fp32 path: %s)' % (name, n, e3, not numpy.array_equal(y3, y)))
            assert e3 < TOL and numpy.isfinite(y3).all()
            for mode in ('bf16x3', 'f32'):
                n2.set_dtype(mode)
                core = engine.VcCore(n1s, n2, mtx)                     # (stage 1 with the pix2pix-init weights: the trained-like stage-1 output is not a mel-cepstrum, exp() of it overflows)
                eff = numpy.ones(n, bool); eff[n // 3:n // 3 + max(1, n // 10)] = False
                mc, spw = core.convert(x[eff], eff)
                core.close()
                mc_ref = numpy.zeros((n, d1.out_ch), numpy.float64); mc_ref[eff] = torch_ref.stage1_convert_core(t1sd, x[eff].astype(numpy.float64))
                mid = (omc.mc2sp(mc_ref, omc.mcepalpha(16000), 1024) + 1e-16).astype(numpy.float32)      # voice_changer.py:38-41: the cast before stage 2
                sp_ref = torch_ref.stage2_convert(t2d, mid.astype(numpy.float64))
                ec = float(numpy.abs(spw.astype(numpy.float64) / sp_ref - 1).max())
                em = float(numpy.abs(mc - mc_ref).max() / numpy.abs(mc_ref).max())
                print('%s chained core n=%d, stage 2 in %s mode: spectrogram element-wise vs float64 oracle %.2e, mc %.2e' % (name, n, mode, ec, em))
                assert ec < TOL and em < TOL and not mc[~eff].any()
    finally:
        n1.close(); n2.close(); n1s.close()


def test_trained_like_statistics_emu(emu_ctx):
    run(emu_ctx, 'SYN-8', 60, [60])


@pytest.mark.gpu
def test_trained_like_statistics_gpu(gpu_ctx):
    """SYN-64, calibrated on a 300-frame window, converted at BASELINE's 300 / 100 / 128-frame windows."""
    run(gpu_ctx, 'SYN-64', 300, [300, 100, 128])
