"""Cross-check of the [MEM] restatements against the REAL upstream packages -- runs only where
`yukarin` / `librosa` are importable from outside this repository (never in the build container, where they are not installable:
every test here then skips).  A maintainer with the real dependencies installed runs this file first."""
import importlib
import sys
from pathlib import Path

import numpy
import pytest

ROOT = Path(__file__).resolve().parent.parent
COMPAT = str(ROOT / 'realtime_yukarin_amd' / 'compat')


def real_module(name):
    """Import `name` with the shim directory taken off sys.path; skip when only the shim (or nothing) is there."""
    saved_path, saved_mods = list(sys.path), {k: v for k, v in sys.modules.items() if k == name or k.startswith(name + '.')}
    for k in saved_mods:
        sys.modules.pop(k)
    sys.path[:] = [p for p in sys.path if str(Path(p).resolve()) not in (COMPAT, str(ROOT / 'tests' / 'stubs'))]
    try:
        mod = importlib.import_module(name)
        if COMPAT in str(getattr(mod, '__file__', '')) or 'tests/stubs' in str(getattr(mod, '__file__', '')):
            pytest.skip('only the shim of %s is importable here' % name)
        return mod
    except ImportError:
        pytest.skip('the real %s is not installed here' % name)
    finally:
        sys.path[:] = saved_path
        for k in [k for k in sys.modules if k == name or k.startswith(name + '.')]:
            sys.modules.pop(k)
        sys.modules.update(saved_mods)


def test_silence_gate_against_librosa():
    librosa = real_module('librosa')
    from realtime_yukarin_amd.compat.yukarin import wave as shim
    rng = numpy.random.default_rng(3)
    w = (0.1 * rng.normal(size=16000)).astype(numpy.float32)
    w[3000:9000] *= 1e-4
    rms = getattr(librosa.feature, 'rms', None) or librosa.feature.rmse
    mse = rms(y=w, frame_length=1024, hop_length=80, center=True, pad_mode='reflect') ** 2
    for thr in (60, 80):
        want = librosa.power_to_db(mse.squeeze()) > -thr
        got = shim.Wave(w, 16000).get_effective_frame(thr, 1024, 5, ref='abs')
        assert numpy.array_equal(got[:len(want)], want[:len(got)])


def test_get_effective_frame_against_yukarin():
    yukarin = real_module('yukarin')
    from realtime_yukarin_amd.compat.yukarin import wave as shim
    rng = numpy.random.default_rng(4)
    w = (0.1 * rng.normal(size=16000)).astype(numpy.float32)
    w[3000:9000] *= 1e-4
    want = yukarin.wave.Wave(wave=w, sampling_rate=16000).get_effective_frame(threshold_db=60, fft_length=1024, frame_period=5)
    for ref in ('abs', 'max'):
        got = shim.Wave(w, 16000).get_effective_frame(60, 1024, 5, ref=ref)
        print('reference level %s: %s' % (ref, 'matches upstream' if numpy.array_equal(got, want) else 'differs'))
    assert numpy.array_equal(shim.Wave(w, 16000).get_effective_frame(60, 1024, 5), want), 'set RY_EFFECTIVE_REF to the form that matches'


def test_silent_feature_and_f0_converter_against_yukarin(tmp_path):
    yukarin = real_module('yukarin')
    from realtime_yukarin_amd.compat.yukarin.acoustic_feature import AcousticFeature as Shim
    sizes = Shim.get_sizes(sampling_rate=16000, order=8)
    real_sizes = yukarin.acoustic_feature.AcousticFeature.get_sizes(sampling_rate=16000, order=8)
    assert {k: sizes[k] for k in ('f0', 'sp', 'ap', 'mc', 'voiced')} == {k: real_sizes[k] for k in ('f0', 'sp', 'ap', 'mc', 'voiced')}
    a = Shim.silent(7, sizes, keys=('mc', 'ap', 'f0', 'voiced'))
    b = yukarin.acoustic_feature.AcousticFeature.silent(7, real_sizes, keys=('mc', 'ap', 'f0', 'voiced'))
    for k in ('mc', 'ap', 'f0', 'voiced'):
        assert numpy.array_equal(numpy.asarray(getattr(a, k)), numpy.asarray(getattr(b, k))), k
    numpy.save(str(tmp_path / 'i.npy'), {'mean': numpy.log(200.0), 'var': 0.04})
    numpy.save(str(tmp_path / 't.npy'), {'mean': numpy.log(300.0), 'var': 0.09})
    from realtime_yukarin_amd.compat.yukarin.f0_converter import F0Converter as ShimF0
    f0 = numpy.array([[0.0], [180.0], [250.0]], numpy.float32)
    want = yukarin.f0_converter.F0Converter(input_statistics=tmp_path / 'i.npy', target_statistics=tmp_path / 't.npy').convert(
        yukarin.acoustic_feature.AcousticFeature(f0=f0.copy())).f0
    got = ShimF0(input_statistics=tmp_path / 'i.npy', target_statistics=tmp_path / 't.npy').convert(f0.copy())
    assert numpy.allclose(got, want, rtol=1e-6)
