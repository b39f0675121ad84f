"""SURVEY.md 8(f) row 2 -- `separate_effective` on the device (`ry_vc_submit_wave`): frame powers, gate, ordered compaction, against
the independent loop-per-frame oracle (oracle/effective_frame.py).  Index / boolean work: the masks must be EQUAL bit for bit, and the
spectrogram must be the one the host-gated path returns for the same mask.  Emulator here, the real GPU under -m gpu."""
import numpy
import pytest

from oracle import effective_frame as oef
from realtime_yukarin_amd import compat, engine, gate, sptk, synth
from realtime_yukarin_amd.weights import flatten_params

compat.install()
FS, FP, HOP = 16000, 5, 80


def waves():
    rng = numpy.random.default_rng(77)
    w = (0.1 * rng.normal(size=300 * HOP)).astype(numpy.float32)
    w[40 * HOP:140 * HOP] *= 1e-4
    w[200 * HOP:230 * HOP] = 0.0
    ramp = (numpy.geomspace(1e-6, 0.5, 120 * HOP) * rng.normal(size=120 * HOP)).astype(numpy.float32)
    w60 = w[:60 * HOP].copy(); w60[10 * HOP:25 * HOP] *= 1e-4; w60[40 * HOP:50 * HOP] = 0.0
    return {'speechlike': w, 'ramp': ramp, 'speech60': w60, 'ramp40': ramp[::3].copy(), 'short': w[:300].copy(), 'tiny': w[:37].copy(), 'one_hop': w[:55].copy(),
            'ragged': w[:1234].copy(), 'all_silent': numpy.zeros(50 * HOP, numpy.float32),
            'all_loud': (0.5 * numpy.sign(rng.normal(size=50 * HOP))).astype(numpy.float32),
            'loud_clamp': (30.0 * rng.normal(size=40 * HOP)).astype(numpy.float32)}          # max power > 20 dB: top_db clamp territory


def make_core(ctx, name, sp_fftlen=256):
    """A small spectrogram (sp_fftlen / 2 + 1 bins) keeps the chain behind the gate cheap: the gate is the thing under test."""
    (d1, P1), (d2, P2) = synth.model_params(name)
    n1 = engine.Net(ctx, d1, flatten_params(d1, P1))
    n2 = engine.Net(ctx, d2, flatten_params(d2, P2), width=sp_fftlen // 2)
    return engine.VcCore(n1, n2, sptk.mc2sp_matrix(8, sptk.mcepalpha(FS), sp_fftlen)), n1, n2


def run_all(ctx, name, fft_lengths=(1024, 256), thrs=(60, 80, 100, 20), deltas=(0, 1, -1), only=None, chain=True):
    core, n1, n2 = make_core(ctx, name)
    rng = numpy.random.default_rng(5)
    checked = 0
    for wname, w in waves().items():
        if only is not None and wname not in only:
            continue
        for fft in fft_lengths:
            for thr in thrs:
                for delta in deltas:
                    n = len(w) // HOP + 1 + delta
                    if n < 1:
                        continue
                    feat = (rng.normal(size=(n, 9)) * synth.MC_SCALE).astype(numpy.float32)
                    want = oef.separate_effective_mask(w, FS, n, thr, fft, FP, 'abs')
                    p_eff, p_all = gate.thresholds(thr)
                    eff, x_eff, rows = core.gate(w, HOP, fft, p_eff, p_all, feat)            # the gate alone: mask, gathered rows, their indices
                    assert numpy.array_equal(eff, want), (wname, fft, thr, delta, numpy.nonzero(eff != want)[0][:8])
                    assert numpy.array_equal(rows, numpy.nonzero(want)[0]) and numpy.array_equal(x_eff, feat[want])
                    if chain and fft == 1024 and thr in (60, 100) and delta == 0:     # the whole window: same as the host-gated call
                        mc, sp, eff2 = core.wait_wave(core.submit_wave(w, HOP, fft, p_eff, p_all, feat))
                        mc2, sp2 = core.convert(feat[want], want)
                        assert numpy.array_equal(eff2, want) and numpy.array_equal(mc, mc2) and numpy.array_equal(sp, sp2)
                        assert not mc[~want].any()
                    checked += 1
    core.close(); n1.close(); n2.close()
    return checked


def test_thresholds_reproduce_the_host_gate_at_the_boundary():
    """The two float32 thresholds ARE the host predicate: one ulp below p_effective the shim's expression says 'not effective'."""
    from yukarin.wave import AMIN
    for thr in (20, 60, 80, 100):
        p_eff, p_all = gate.thresholds(thr)
        for p, want in ((numpy.float32(p_eff), True), (numpy.nextafter(numpy.float32(p_eff), numpy.float32(0)), False)):
            db = 10.0 * numpy.log10(numpy.maximum(AMIN, numpy.full(7, p, numpy.float32)))
            assert bool((db > -thr).all()) == want
        assert p_all > p_eff


def test_device_gate_masks_are_bit_equal_emu(emu_ctx):
    # (the emulator runs a 1024-fiber workgroup per window: a subset here, the full cross product on the GPU)
    assert run_all(emu_ctx, 'SYN-8', only=('speech60', 'short', 'one_hop', 'ragged', 'all_silent', 'loud_clamp'),
                   chain=False) >= 130
    assert run_all(emu_ctx, 'SYN-8', fft_lengths=(1024,), thrs=(60,), deltas=(0,), only=('speech60', 'tiny', 'all_silent')) == 3


def check_the_clamp_is_decided_by_every_wave_frame(ctx, name):
    """The `top_db` clamp of librosa.power_to_db takes the maximum over ALL len(wave) // hop + 1 wave frames; the feature block of a live
    window is one frame shorter (n * hop samples -> n + 1 wave frames), so the loudest frame can be the one past the block.  A silent wave
    with a loud tail, threshold 100 dB: the tail lifts max - 80 dB above -100 dB, every frame passes on the host -- and must on the device
    (round-2 advisor: the device clipped the frame count before taking the maximum and returned no frame)."""
    core, n1, n2 = make_core(ctx, name)
    rng = numpy.random.default_rng(6)
    cases = 0
    for fft in (128, 1024):
        for tail in (16, 3):
            w = numpy.zeros(50 * HOP, numpy.float32); w[-tail:] = 0.5
            for delta in (-1, 0, -7):
                n = len(w) // HOP + 1 + delta
                feat = (rng.normal(size=(n, 9)) * synth.MC_SCALE).astype(numpy.float32)
                for thr in (100, 60):
                    want = oef.separate_effective_mask(w, FS, n, thr, fft, FP, 'abs')
                    p_eff, p_all = gate.thresholds(thr)
                    eff, x_eff, rows = core.gate(w, HOP, fft, p_eff, p_all, feat)
                    assert numpy.array_equal(eff, want), (fft, tail, delta, thr, int(eff.sum()), int(want.sum()))
                    assert numpy.array_equal(x_eff, feat[want])
                    if thr == 100 and delta == -1 and fft == 128 and tail == 16:
                        assert want.all(), 'the case is the one the advisor reproduced: 50 of 50 frames on the host'
                    cases += 1
    # a window that grows: the wave staging of the slot is replaced, not leaked; results unaffected
    w2 = (0.1 * rng.normal(size=400 * HOP)).astype(numpy.float32)
    feat2 = (rng.normal(size=(400, 9)) * synth.MC_SCALE).astype(numpy.float32)
    p_eff, p_all = gate.thresholds(60)
    eff, _, _ = core.gate(w2, HOP, 1024, p_eff, p_all, feat2)
    assert numpy.array_equal(eff, oef.separate_effective_mask(w2, FS, 400, 60, 1024, FP, 'abs'))
    core.close(); n1.close(); n2.close()
    return cases


def test_the_clamp_is_decided_by_every_wave_frame_emu(emu_ctx):
    assert check_the_clamp_is_decided_by_every_wave_frame(emu_ctx, 'SYN-8') == 24


@pytest.mark.gpu
def test_the_clamp_is_decided_by_every_wave_frame_gpu(gpu_ctx):
    assert check_the_clamp_is_decided_by_every_wave_frame(gpu_ctx, 'SYN-8') == 24


def test_mirror_voice_changer_takes_the_device_gate(emu_ctx, monkeypatch, tmp_path):
    """The mirror VoiceChanger uses `submit_wave` for a float32 wave and falls back to the host gate otherwise; same outputs."""
    import test_shims_e2e as e2e
    monkeypatch.setattr(engine, 'get_context', lambda device=0, lib=None: emu_ctx)
    from realtime_yukarin_amd.voice_changer import VoiceChanger
    from yukarin import AcousticFeature, Wave
    e2e.write_models(tmp_path, 'SYN-8')
    ac, sr = e2e.build_converters(tmp_path)
    wave, feat = e2e.make_window(60, 21)
    calls = []
    real = engine.VcCore.submit_wave
    monkeypatch.setattr(engine.VcCore, 'submit_wave', lambda self, *a, **k: calls.append(1) or real(self, *a, **k))

    def run(w):
        class Wrapped(AcousticFeature):
            pass
        f = Wrapped(**{k: v.copy() for k, v in feat.items()}); f.wave = Wave(wave=w, sampling_rate=FS)
        return VoiceChanger(acoustic_converter=ac, super_resolution=sr, threshold=60).convert_from_acoustic_feature(f)
    dev = run(wave)
    assert calls == [1]
    host = run(wave.astype(numpy.float64).astype(numpy.float32).astype(numpy.float64))     # a float64 wave: host gate (different arithmetic)
    assert calls == [1]
    monkeypatch.setenv('RY_DEVICE_GATE', '0')
    host32 = run(wave)
    assert calls == [1]
    for o in (host32,):
        assert numpy.array_equal(dev.sp, o.sp) and numpy.array_equal(dev.mc, o.mc) and numpy.array_equal(dev.f0, o.f0) and numpy.array_equal(dev.ap, o.ap)
    assert float(numpy.abs(dev.sp / host.sp - 1).max()) < 1e-5


def _long_window(ctx):
    """More than 1024 frames: the compaction workgroup walks the window in chunks of 1024 and carries the running count."""
    core, n1, n2 = make_core(ctx, 'SYN-8')
    rng = numpy.random.default_rng(9)
    n = 2500
    w = (0.1 * rng.normal(size=n * HOP)).astype(numpy.float32)
    for a, b in ((100, 400), (1000, 1100), (1500, 2300)):
        w[a * HOP:b * HOP] *= 1e-5
    feat = rng.normal(size=(n + 1, 9)).astype(numpy.float32)
    want = oef.separate_effective_mask(w, FS, n + 1, 60, 1024, FP, 'abs')
    eff, x_eff, rows = core.gate(w, HOP, 1024, *gate.thresholds(60), feat)
    assert numpy.array_equal(eff, want) and numpy.array_equal(rows, numpy.nonzero(want)[0]) and numpy.array_equal(x_eff, feat[want])
    assert 1000 < want.sum() < 2000
    core.close(); n1.close(); n2.close()


def test_device_gate_long_window_emu(emu_ctx):
    _long_window(emu_ctx)


@pytest.mark.gpu
def test_device_gate_long_window_gpu(gpu_ctx):
    _long_window(gpu_ctx)


def _fixtures_against(ctx):
    import glob
    from pathlib import Path
    core, n1, n2 = make_core(ctx, 'SYN-8')
    files = sorted(glob.glob(str(Path(__file__).resolve().parent / 'golden' / 'gate' / '*.npz')))
    checked = 0
    for f in files:
        z = numpy.load(f)
        n = int(z['n_frames'])
        feat = numpy.zeros((n, 9), numpy.float32)
        for key in z.files:
            if not key.startswith('abs_'):
                continue                                           # the device gate restates the absolute form
            _, thr, fft = key.split('_')
            eff, _, rows = core.gate(z['wave'], HOP, int(fft[3:]), *gate.thresholds(int(thr[3:])), feat)
            assert numpy.array_equal(eff, z[key]) and numpy.array_equal(rows, numpy.nonzero(z[key])[0]), (f, key)
            checked += 1
    core.close(); n1.close(); n2.close()
    return checked


def test_device_gate_matches_the_committed_fixtures_emu(emu_ctx):
    assert _fixtures_against(emu_ctx) == 40


@pytest.mark.gpu
def test_device_gate_matches_the_committed_fixtures_gpu(gpu_ctx):
    assert _fixtures_against(gpu_ctx) == 40


@pytest.mark.gpu
def test_device_gate_masks_are_bit_equal_gpu(gpu_ctx):
    assert run_all(gpu_ctx, 'SYN-8') > 160
