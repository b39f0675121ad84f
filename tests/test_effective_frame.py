"""A1 -- the silence gate (`separate_effective`, /root/reference/realtime_voice_conversion/yukarin_wrapper/voice_changer.py:27-31).
The shim's vectorised restatement of the librosa call sequence (compat/yukarin/wave.py) against the independent loop-per-frame
restatement in oracle/effective_frame.py: index / boolean work, so the masks must be EQUAL, bit for bit."""
import numpy
import pytest

from oracle import effective_frame as oef
from realtime_yukarin_amd import compat

compat.install()
FS, FP, FFT = 16000, 5, 1024
HOP = FS * FP // 1000


def shim_mask(wave, thr, ref):
    from yukarin import Wave
    return Wave(wave=wave, sampling_rate=FS).get_effective_frame(threshold_db=thr, fft_length=FFT, frame_period=FP, ref=ref)


def waves():
    rng = numpy.random.default_rng(77)
    w = (0.1 * rng.normal(size=300 * HOP)).astype(numpy.float32)
    w[40 * HOP:140 * HOP] *= 1e-4                       # a quiet stretch around -100 dB
    w[200 * HOP:230 * HOP] = 0.0                        # digital silence
    ramp = (numpy.geomspace(1e-6, 0.5, 120 * HOP) * rng.normal(size=120 * HOP)).astype(numpy.float32)   # crosses every threshold
    return {
        'speechlike_300_frames': w,
        'ramp_through_the_thresholds': ramp,
        'float64_wave': w[:60 * HOP].astype(numpy.float64),
        'shorter_than_fft': w[:300].copy(),             # 300 samples < fft_length: reflect padding wraps around more than once
        'shorter_than_half_fft': w[:37].copy(),
        'shorter_than_one_hop': w[:55].copy(),          # a single frame: numpy reduces a (fft, 1) block
        'one_sample': w[:1].copy(),
        'not_a_multiple_of_hop': w[:1234].copy(),
        'all_silent': numpy.zeros(50 * HOP, numpy.float32),
        'all_loud': (0.5 * numpy.sign(rng.normal(size=50 * HOP))).astype(numpy.float32),
        'int16_wave': (w[:40 * HOP] * 32767).astype(numpy.int16),
    }


@pytest.mark.parametrize('name', sorted(waves()))
@pytest.mark.parametrize('ref', ['abs', 'max'])
@pytest.mark.parametrize('thr', [60, 80, 100, 20])
def test_masks_are_bit_equal(name, ref, thr):
    w = waves()[name]
    got = shim_mask(w, thr, ref)
    want = oef.effective_frames(w, FS, thr, FFT, FP, ref)
    assert got.dtype == numpy.bool_ and got.shape == (len(w) // HOP + 1,)
    assert numpy.array_equal(got, want), (name, ref, thr, numpy.nonzero(got != want)[0][:8])


def test_the_gate_does_something_on_the_designed_waves():
    w = waves()
    m = shim_mask(w['speechlike_300_frames'], 80, 'abs')
    assert 0 < m.sum() < len(m) and not m[208:222].any() and m[:30].all() and not m[60:120].any()
    assert not shim_mask(w['all_silent'], 80, 'abs').any()                 # absolute gate: digital silence has no effective frame
    assert shim_mask(w['all_silent'], 80, 'max').all()                     # relative gate: 0 dB below its own maximum everywhere
    assert shim_mask(w['all_loud'], 80, 'abs').all()
    r = shim_mask(w['ramp_through_the_thresholds'], 60, 'abs')
    assert not r[0] and r[-1] and (numpy.diff(r.astype(int)) != 0).sum() >= 1


def test_empty_wave():
    assert shim_mask(numpy.zeros(0, numpy.float32), 60, 'abs').shape == (0,)
    assert oef.effective_frames(numpy.zeros(0, numpy.float32), FS, 60, FFT, FP).shape == (0,)


def test_pairwise_model_is_numpys():
    """Pins oracle.effective_frame.pairwise_sum to what numpy's float reduction does on the layout librosa's frame view produces."""
    rng = numpy.random.default_rng(1)
    for fl, hop, nf in ((1024, 80, 40), (1024, 120, 7), (512, 80, 3), (1024, 80, 1), (64, 16, 50), (1000, 80, 11), (7, 3, 9)):
        y = rng.normal(size=fl + hop * (nf - 1) + 5).astype(numpy.float32)
        it = y.strides[0]
        v = numpy.lib.stride_tricks.as_strided(y, shape=(fl, nf), strides=(it, hop * it))
        sq = numpy.abs(v) ** 2
        m = numpy.mean(sq, axis=0)
        mine = numpy.array([numpy.float32(oef.pairwise_sum(list(sq[:, j]), 0, fl, numpy.float32) / numpy.float32(fl)) for j in range(nf)])
        assert numpy.array_equal(m, mine), (fl, hop, nf)


@pytest.mark.parametrize('ref', ['abs', 'max'])
@pytest.mark.parametrize('delta', [-1, 0, 1, 2])
@pytest.mark.parametrize('thr', [60, None])
def test_separate_effective_against_the_oracle_with_a_frame_count_mismatch(tmp_path, monkeypatch, ref, delta, thr):
    """`AcousticConverter.separate_effective`: the wave gives len // hop + 1 frames, WORLD features one more or fewer."""
    from yukarin import AcousticFeature, Wave
    from yukarin.acoustic_converter import AcousticConverter
    monkeypatch.setenv('RY_EFFECTIVE_REF', ref)
    w = waves()['speechlike_300_frames'][:120 * HOP]
    n = len(w) // HOP + 1 + delta
    rng = numpy.random.default_rng(5)
    feat = AcousticFeature(f0=rng.random((n, 1)).astype('f4'), mc=rng.normal(size=(n, 9)).astype('f4'),
                           ap=rng.random((n, 513)).astype('f4'), voiced=rng.random((n, 1)) > 0.5)

    class P(object):
        sampling_rate, frame_period, fft_length, threshold_db = FS, FP, FFT, None
    ac = AcousticConverter.__new__(AcousticConverter)
    ac._param = P()
    f_eff, eff = ac.separate_effective(wave=Wave(wave=w, sampling_rate=FS), feature=feat, threshold=thr)
    want = oef.separate_effective_mask(w, FS, n, thr, FFT, FP, ref)
    assert eff.dtype == numpy.bool_ and numpy.array_equal(eff, want)
    assert numpy.array_equal(f_eff.mc, feat.mc[want]) and numpy.array_equal(f_eff.ap, feat.ap[want]) and len(f_eff.f0) == int(want.sum())
    if thr is None:
        assert eff.all()


def test_bad_reference_name_is_refused(monkeypatch):
    monkeypatch.setenv('RY_EFFECTIVE_REF', 'peak')
    with pytest.raises(ValueError, match='RY_EFFECTIVE_REF'):
        shim_mask(numpy.ones(400, numpy.float32), 60, None)


def test_shim_matches_the_committed_gate_fixtures():
    """tests/golden/gate/*.npz (made by tests/golden/make_gate_golden.py from the oracle): a committed target independent of the oracle code."""
    import glob
    from pathlib import Path
    files = sorted(glob.glob(str(Path(__file__).resolve().parent / 'golden' / 'gate' / '*.npz')))
    assert len(files) == 5
    for f in files:
        z = numpy.load(f)
        for key in z.files:
            if key in ('wave', 'n_frames'):
                continue
            ref, thr, fft = key.split('_')
            got = shim_mask_fft(z['wave'], int(thr[3:]), ref, int(fft[3:]))
            assert numpy.array_equal(got[:int(z['n_frames'])], z[key]), (f, key)


def shim_mask_fft(wave, thr, ref, fft):
    from yukarin import Wave
    return Wave(wave=wave, sampling_rate=FS).get_effective_frame(threshold_db=thr, fft_length=fft, frame_period=FP, ref=ref)
