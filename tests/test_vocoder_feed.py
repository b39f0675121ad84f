"""`realtime_yukarin_amd.vocoder_feed.decode` against the reference's `RealtimeVocoder.decode`
(/root/reference/realtime_voice_conversion/yukarin_wrapper/vocoder.py:88-117) on a recording stand-in of world4py."""
import importlib
import sys
import time
from pathlib import Path

import numpy
import pytest

import world4py_fake
from realtime_yukarin_amd import compat, vocoder_feed

compat.install()
ROOT = Path(__file__).resolve().parent.parent
REF = Path('/root/reference')


def feature(n, seed, dtype=numpy.float32):
    from yukarin import AcousticFeature
    rng = numpy.random.default_rng(seed)
    f0 = numpy.where(rng.random((n, 1)) < 0.3, 0.0, rng.lognormal(numpy.log(220.0), 0.2, (n, 1)))
    return AcousticFeature(f0=f0.astype(dtype), sp=numpy.exp(rng.normal(-6, 1.5, (n, 513))).astype(dtype),
                           ap=rng.uniform(0.001, 0.999, (n, 513)).astype(dtype), voiced=f0 > 0)


class Holder(object):
    """The attributes `decode` touches on a RealtimeVocoder."""
    def __init__(self):
        self._synthesizer = world4py_fake.WorldSynthesizer()
        self._before_buffer = []
        self.out_sampling_rate = 24000


def test_feed_pointers_address_the_values():
    import ctypes
    f = feature(7, 1)
    feed = vocoder_feed.Feed(f.f0, f.sp[:, ::2], f.ap)                       # a non-contiguous sp is copied, not mis-addressed
    assert feed.length == 7 and [feed.f0_pointer[i] for i in range(7)] == [float(v) for v in f.f0[:, 0]]
    for i in (0, 3, 6):
        assert feed.sp_pointer[i][5] == float(f.sp[i, 10]) and feed.ap_pointer[i][512] == float(f.ap[i, 512])
    with pytest.raises(ValueError):
        vocoder_feed.Feed(f.f0, f.sp[:5], f.ap)


def test_decode_matches_the_list_based_feed_on_the_recording_world4py(monkeypatch):
    world4py_fake.install(monkeypatch)
    a, b = Holder(), Holder()
    for seed, n in ((1, 20), (2, 21), (3, 1), (4, 40)):                        # odd counts, a window too short for a block
        f = feature(n, seed)
        got = vocoder_feed.decode(a, f)
        # the reference's way, restated: lists -> world4py casts -> per-sample read back
        f0_buffer = world4py_fake.cast_1d_list_to_1d_pointer(f.f0.flatten().tolist())
        sp_buffer = world4py_fake.cast_2d_list_to_2d_pointer(f.sp.tolist())
        ap_buffer = world4py_fake.cast_2d_list_to_2d_pointer(f.ap.tolist())
        world4py_fake._AddParameters(f0_buffer, n, sp_buffer, ap_buffer, b._synthesizer)
        ys = []
        while world4py_fake._Synthesis2(b._synthesizer) != 0:
            ys.append(numpy.array([b._synthesizer.buffer[i] for i in range(b._synthesizer.buffer_size)]))
        want = numpy.concatenate(ys) if ys else numpy.empty(0)
        assert got.sampling_rate == 24000 and got.wave.dtype == numpy.float64
        assert numpy.array_equal(got.wave, want) and len(got.wave) == 64 * (n // 2)
        assert a._synthesizer.calls == b._synthesizer.calls
    for _ in range(20):
        vocoder_feed.decode(a, feature(4, 9))
    assert len(a._before_buffer) == 16                                           # the same keep-alive window as the reference


@pytest.mark.skipif(not REF.exists(), reason='/root/reference is not mounted here')
def test_bound_over_the_reference_class_it_gives_the_reference_result(monkeypatch):
    """`RealtimeVocoder.decode = vocoder_feed.decode`: same waves as the reference's own method, and faster."""
    for p in (str(ROOT / 'tests' / 'stubs'), str(REF)):
        monkeypatch.syspath_prepend(p)
    world4py_fake.install(monkeypatch)
    for m in [k for k in sys.modules if k.startswith('realtime_voice_conversion')]:      # an earlier test may have imported it over the stub world4py
        sys.modules.pop(m)
    voc = importlib.import_module('realtime_voice_conversion.yukarin_wrapper.vocoder')

    def make():
        v = voc.RealtimeVocoder.__new__(voc.RealtimeVocoder)
        v._synthesizer = world4py_fake.WorldSynthesizer()
        v._before_buffer = []
        v.out_sampling_rate = 24000
        return v
    ref, fast = make(), make()
    feats = [feature(100, s, numpy.float64) for s in range(4)]                   # decode side is float64 (vocoder.py:54)
    t0 = time.perf_counter(); want = [voc.RealtimeVocoder.decode(ref, f) for f in feats]; t_ref = time.perf_counter() - t0
    monkeypatch.setattr(voc.RealtimeVocoder, 'decode', vocoder_feed.decode)
    t0 = time.perf_counter(); got = [fast.decode(f) for f in feats]; t_fast = time.perf_counter() - t0
    for g, w in zip(got, want):
        assert numpy.array_equal(g.wave, w.wave) and g.sampling_rate == w.sampling_rate
    print('RealtimeVocoder.decode feed, 4 x 100 frames: reference %.1f ms, pointer feed %.1f ms' % (t_ref * 1e3, t_fast * 1e3))
    for m in [k for k in sys.modules if k.startswith('realtime_voice_conversion')]:
        sys.modules.pop(m)
