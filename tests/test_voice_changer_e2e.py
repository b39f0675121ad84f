"""End-to-end drop-in check of the hot-path entry on CPU (emulator build of the kernels):
`VoiceChanger.convert_from_acoustic_feature` driven through the `yukarin` / `become_yukarin` shims, with model files
written to disk in the Chainer save_npz key layout + config.json, compared against the oracle CNNs composed with the
same host steps.  Runs the reference's OWN VoiceChanger class when /root/reference is mounted, and always the
host-side mirror `realtime_yukarin_amd.voice_changer.VoiceChanger`."""
import importlib
import json
import pickle
import sys
from pathlib import Path

import numpy
import pytest

from oracle import effective_frame as oef
from oracle import mc2sp as omc
from oracle import unet
from realtime_yukarin_amd import compat, engine, sptk
from realtime_yukarin_amd.netspec import NetDesc
from realtime_yukarin_amd.weights import save_npz, synthetic_params

compat.install()
ROOT = Path(__file__).resolve().parent.parent
REF = Path('/root/reference')
FS, FRAME_PERIOD, N = 16000, 5, 60


@pytest.fixture(scope='module')
def models(tmp_path_factory):
    d = tmp_path_factory.mktemp('models')
    d1, d2 = NetDesc(1, 9, 9, 8, 8), NetDesc(2, 1, 1, 8, 8)
    P1, P2 = synthetic_params(d1, 11, bias_std=0.05), synthetic_params(d2, 12, bias_std=0.02)
    P1n = dict(P1); P1n['encoder/c1/batchnorm/N'] = numpy.array(7)          # Chainer also stores the BN sample counter
    save_npz(d / 's1.npz', P1n)
    save_npz(d / 's2.npz', P2)
    (d / 's1.json').write_text(json.dumps({
        'dataset': {'acoustic_param': {'sampling_rate': FS, 'frame_period': FRAME_PERIOD, 'order': 8, 'alpha': 0.41, 'unknown_key': 1},
                    'in_features': ['mc'], 'out_features': ['mc'], 'input_glob': 'x'},
        'model': {'in_channels': 9, 'out_channels': 9, 'generator_base_channels': 8, 'generator_extensive_layers': 8,
                  'discriminator_base_channels': 1}, 'loss': {}, 'train': {}}))
    (d / 's2.json').write_text(json.dumps({
        'dataset': {'param': {'voice_param': {'sample_rate': FS}, 'acoustic_feature_param': {'frame_period': FRAME_PERIOD, 'order': 8}}},
        'model': {'generator_base_channels': 8, 'generator_extensive_layers': 8}}))
    numpy.save(str(d / 'in_stat.npy'), {'mean': numpy.log(200.0), 'var': 0.04})
    numpy.save(str(d / 'tg_stat.npy'), {'mean': numpy.log(300.0), 'var': 0.09})
    return d, (d1, P1), (d2, P2)


@pytest.fixture()
def on_emulator(emu_ctx, monkeypatch):
    """Route the shims' lazily created context to the emulator build (tests only; the product never does this)."""
    monkeypatch.setattr(engine, 'get_context', lambda device=0, lib=None: emu_ctx)
    return emu_ctx


def make_input(rng):
    from yukarin import Wave
    wave = (0.1 * rng.normal(size=N * FS * FRAME_PERIOD // 1000)).astype(numpy.float32)
    wave[10 * 80:40 * 80] = 0.0                                                 # a silent stretch -> separate_effective drops frames
    f0 = numpy.where(rng.random((N, 1)) < 0.3, 0.0, rng.lognormal(numpy.log(220.0), 0.2, (N, 1))).astype(numpy.float32)
    feat = dict(f0=f0, ap=rng.uniform(0.001, 0.999, (N, 513)).astype(numpy.float32),
                mc=(rng.normal(size=(N, 9)) * [4, 1, .5, .5, .3, .3, .2, .2, .2]).astype(numpy.float32), voiced=f0 > 0)
    return Wave(wave=wave, sampling_rate=FS), feat


def build_converters(models):
    from become_yukarin import SuperResolution
    from become_yukarin.config.sr_config import create_from_json as create_sr_config
    from yukarin import AcousticConverter
    from yukarin.config import create_from_json as create_config
    from yukarin.f0_converter import F0Converter
    d = models[0]
    f0c = F0Converter(input_statistics=d / 'in_stat.npy', target_statistics=d / 'tg_stat.npy')
    ac = AcousticConverter(create_config(d / 's1.json'), d / 's1.npz', f0_converter=f0c, out_sampling_rate=FS)
    sr = SuperResolution(create_sr_config(d / 's2.json'), d / 's2.npz')
    return ac, sr


def expected(models, ac, wave, feat):
    """Oracle CNNs + the same host glue, written out step by step (voice_changer.py:24-42)."""
    (_, P1), (_, P2) = models[1], models[2]
    eff = oef.separate_effective_mask(wave.wave, FS, N, 60, 1024, FRAME_PERIOD)      # the independent loop-per-frame gate (A1)
    mc = numpy.zeros((N, 9), numpy.float32)
    mc[eff] = unet.stage1_convert_core(feat['mc'][eff], P1)
    f0 = numpy.zeros((N, 1), numpy.float32)
    f0[eff] = ac.f0_converter.convert(feat['f0'][eff])
    sp = omc.mc2sp(mc, omc.mcepalpha(FS), 1024) + 1e-16
    return dict(f0=f0, mc=mc, sp=unet.stage2_convert(sp.astype(numpy.float32), P2), eff=eff)


def check(out, exp, feat):
    assert out.sp.shape == (N, 513) and out.sp.dtype == numpy.float32
    assert float(numpy.abs(out.sp / exp['sp'] - 1).max()) < 1e-4
    assert float(numpy.abs(out.mc - exp['mc']).max() / numpy.abs(exp['mc']).max()) < 1e-4
    assert numpy.allclose(out.f0, exp['f0'], rtol=1e-6)
    assert numpy.array_equal(out.ap[exp['eff']], feat['ap'][exp['eff']]) and not out.ap[~exp['eff']].any()
    assert 0 < exp['eff'].sum() < N


def test_device_mc2sp_matches_the_sptk_recursion(emu_ctx):
    """`decode_spectrogram`: exp(mc @ M) on the device vs the independent scalar-loop restatement of pysptk.mc2sp (oracle/mc2sp.py)."""
    rng = numpy.random.default_rng(3)
    mc = (rng.normal(size=(37, 9)) * [4, 1, .5, .5, .3, .3, .2, .2, .2]).astype(numpy.float32)
    alpha = sptk.mcepalpha(FS)
    ref = omc.mc2sp_sptk(mc, omc.mcepalpha(FS), 1024)
    got = emu_ctx.mc2sp(mc, sptk.mc2sp_matrix(8, alpha, 1024))
    assert got.shape == (37, 513) and float(numpy.abs(got / ref - 1).max()) < 5e-5
    assert float(numpy.abs(sptk.mc2sp_fast(mc, alpha, 1024) / ref - 1).max()) < 1e-12


def test_mirror_voice_changer_end_to_end(models, on_emulator):
    from realtime_yukarin_amd.voice_changer import VoiceChanger
    from yukarin import AcousticFeature
    ac, sr = build_converters(models)
    ac = pickle.loads(pickle.dumps(ac)); sr = pickle.loads(pickle.dumps(sr))     # run.py ships them to a child Process
    wave, feat = make_input(numpy.random.default_rng(21))

    class Wrapped(AcousticFeature):                                              # AcousticFeatureWrapper equivalent
        pass
    f_in = Wrapped(**{k: v.copy() for k, v in feat.items()}); f_in.wave = wave
    vc = VoiceChanger(acoustic_converter=ac, super_resolution=sr, threshold=60)
    assert vc.output_sampling_rate == FS
    out = vc.convert_from_acoustic_feature(f_in)
    check(out, expected(models, ac, wave, feat), feat)
    # the generic (step-by-step) path and the device-resident fused path agree
    assert vc._fused_core() is not None
    vc2 = VoiceChanger(acoustic_converter=ac, super_resolution=sr, threshold=60)
    vc2._fused_core = lambda: None
    f_gen = Wrapped(**{k: v.copy() for k, v in feat.items()}); f_gen.wave = wave
    gen = vc2.convert_from_acoustic_feature(f_gen)
    assert float(numpy.abs(gen.sp / out.sp - 1).max()) < 2e-5 and numpy.allclose(gen.mc, out.mc, rtol=1e-6, atol=1e-7)
    assert numpy.array_equal(gen.f0, out.f0) and numpy.array_equal(gen.ap, out.ap) and numpy.array_equal(gen.voiced, out.voiced)
    # no effective frame at all: the stage-1 CNN is skipped (voice_changer.py:32-35): mc = 0, sp = SuperResolution(exp(0) + 1e-16)
    mc0, sp0 = vc._fused_core().convert(numpy.zeros((0, 9), numpy.float32), numpy.zeros(N, bool))
    assert not mc0.any() and sp0.shape == (N, 513)
    ones = numpy.ones((N, 513), numpy.float32)
    assert float(numpy.abs(sp0 / unet.stage2_convert(ones, models[2][1]) - 1).max()) < 1e-4
    # two windows in one batched stage-2 call give the same answer
    f_a = Wrapped(**{k: v.copy() for k, v in feat.items()}); f_a.wave = wave
    f_b = Wrapped(**{k: v.copy() for k, v in feat.items()}); f_b.wave = wave
    both = vc.convert_windows([f_a, f_b])
    assert numpy.allclose(both[0].sp, out.sp, rtol=1e-6) and numpy.allclose(both[1].sp, out.sp, rtol=1e-6)


def test_all_silent_window_skips_the_stage1_cnn_through_the_wave_gate(models, on_emulator, monkeypatch):
    """voice_changer.py:32-35: no effective frame -> `f_out = f_in_effective` (the CNN is not called), mc stays all-silent zeros,
    sp = SuperResolution(mc2sp(0) + 1e-16).  Reached through the wave gate itself (an all-zero and a very quiet window), on the
    mirror class in both its fused and step-by-step forms."""
    from realtime_yukarin_amd.voice_changer import VoiceChanger
    from yukarin import AcousticFeature, Wave
    ac, sr = build_converters(models)
    rng = numpy.random.default_rng(8)
    _, feat = make_input(rng)
    calls = []
    real_convert = type(ac).convert
    monkeypatch.setattr(type(ac), 'convert', lambda self, f: calls.append(len(f.f0)) or real_convert(self, f))
    want_sp = unet.stage2_convert(numpy.ones((N, 513), numpy.float32), models[2][1])

    class Wrapped(AcousticFeature):
        pass
    for amp in (0.0, 3e-6):                                                   # digital silence / -110 dB noise: below the 60 dB gate
        wave = Wave(wave=(amp * rng.normal(size=N * FS * FRAME_PERIOD // 1000)).astype(numpy.float32), sampling_rate=FS)
        assert not oef.separate_effective_mask(wave.wave, FS, N, 60, 1024, FRAME_PERIOD).any()
        for fused in (True, False):
            vc = VoiceChanger(acoustic_converter=ac, super_resolution=sr, threshold=60)
            if not fused:
                vc._fused_core = lambda: None
            f_in = Wrapped(**{k: v.copy() for k, v in feat.items()}); f_in.wave = wave
            out = vc.convert_from_acoustic_feature(f_in)
            assert out.mc.shape == (N, 9) and not out.mc.any() and not out.f0.any() and not out.ap.any() and not out.voiced.any()
            assert float(numpy.abs(out.sp / want_sp - 1).max()) < 1e-4
    assert calls == []                                                        # AcousticConverter.convert never ran


@pytest.mark.skipif(not REF.exists(), reason='/root/reference is not mounted here')
def test_super_resolution_dtype_is_selectable_from_the_environment(models, on_emulator, monkeypatch):
    """RY_SR_DTYPE lets an unchanged run.py / check.py opt into the split-bf16 (or bf16) stage-2 arithmetic; anything else
    is refused before the first convert.  (The base-8 test model has no MFMA-bound layer, so the result is the fp32 one.)"""
    rng = numpy.random.default_rng(5)
    sp = numpy.exp(rng.normal(-6.0, 1.5, (40, 513))).astype(numpy.float32)
    _, sr = build_converters(models)
    ref = sr.convert(sp)
    monkeypatch.setenv('RY_SR_DTYPE', 'bf16x3')
    _, sr3 = build_converters(models)
    assert float(numpy.abs(sr3.convert(sp) / ref - 1).max()) < 1e-4
    monkeypatch.setenv('RY_SR_DTYPE', 'fp16')
    _, bad = build_converters(models)
    with pytest.raises(ValueError, match='RY_SR_DTYPE'):
        bad.convert(sp)


def test_reference_voice_changer_and_convert_stream_run_unchanged_on_the_shims(models, on_emulator, monkeypatch):
    for p in (str(ROOT / 'tests' / 'stubs'), str(REF)):
        monkeypatch.syspath_prepend(p)
    vc_mod = importlib.import_module('realtime_voice_conversion.yukarin_wrapper.voice_changer')
    cs_mod = importlib.import_module('realtime_voice_conversion.stream.convert_stream')
    ac, sr = build_converters(models)
    wave, feat = make_input(numpy.random.default_rng(21))
    f_in = vc_mod.AcousticFeatureWrapper(wave=wave, **{k: v.copy() for k, v in feat.items()})
    vc = vc_mod.VoiceChanger(acoustic_converter=ac, super_resolution=sr, threshold=60)
    out = vc.convert_from_acoustic_feature(f_in)
    check(out, expected(models, ac, wave, feat), feat)
    # ... and the reference's ConvertStream drives it with overlap windows (time 0.3 s = 60 frames, no extra)
    stream = cs_mod.ConvertStream(voice_changer=vc)
    stream.add(start_time=0, data=vc_mod.AcousticFeatureWrapper(wave=wave, **{k: v.copy() for k, v in feat.items()}))
    got = stream.process(start_time=0, time_length=0.3, extra_time=0)
    assert numpy.allclose(got.sp, out.sp, rtol=1e-6)
    for m in [k for k in sys.modules if k.startswith('realtime_voice_conversion')]:
        sys.modules.pop(m)


@pytest.mark.skipif(not REF.exists(), reason='/root/reference is not mounted here')
def test_unchanged_reference_class_keeps_the_data_on_the_device_between_the_cnns(models, on_emulator, monkeypatch):
    """The reference's OWN VoiceChanger, not edited, calling the shims step by step (voice_changer.py:33-41): `convert` leaves its
    rows on the device, `decode_spectrogram` hands back a lazy spectrogram, `+= 1e-16` / `astype` are remembered and
    `SuperResolution.convert` continues on the device -- one small H2D, no second upload, no host mc2sp -- with the same
    outputs as the plain step-by-step path (RY_FUSE_STEPS=0)."""
    from realtime_yukarin_amd import fusion
    for p in (str(ROOT / 'tests' / 'stubs'), str(REF)):
        monkeypatch.syspath_prepend(p)
    vc_mod = importlib.import_module('realtime_voice_conversion.yukarin_wrapper.voice_changer')
    ac, sr = build_converters(models)
    wave, feat = make_input(numpy.random.default_rng(21))
    calls = {'stage1': 0, 'stage2_from_mc': 0, 'mid_sp': 0, 'net_convert': 0, 'host_mc2sp': 0}
    for name, key in (('convert_stage1', 'stage1'), ('stage2_from_mc', 'stage2_from_mc'), ('mid_sp', 'mid_sp')):
        real = getattr(engine.VcCore, name)
        monkeypatch.setattr(engine.VcCore, name, (lambda real, key: lambda self, *a, **k: calls.__setitem__(key, calls[key] + 1) or real(self, *a, **k))(real, key))
    real_nc = engine.Net.convert
    monkeypatch.setattr(engine.Net, 'convert', lambda self, x, discard=(0, 0): calls.__setitem__('net_convert', calls['net_convert'] + 1) or real_nc(self, x, discard))
    real_fast = sptk.mc2sp_fast
    monkeypatch.setattr(sptk, 'mc2sp_fast', lambda *a, **k: calls.__setitem__('host_mc2sp', calls['host_mc2sp'] + 1) or real_fast(*a, **k))

    def run():
        f_in = vc_mod.AcousticFeatureWrapper(wave=wave, **{k: v.copy() for k, v in feat.items()})
        return vc_mod.VoiceChanger(acoustic_converter=ac, super_resolution=sr, threshold=60).convert_from_acoustic_feature(f_in)
    out = run()
    assert calls == {'stage1': 1, 'stage2_from_mc': 1, 'mid_sp': 0, 'net_convert': 0, 'host_mc2sp': 0}, calls
    check(out, expected(models, ac, wave, feat), feat)
    assert isinstance(out.sp, numpy.ndarray) and out.sp.dtype == numpy.float32
    monkeypatch.setenv('RY_FUSE_STEPS', '0')
    plain = run()
    assert calls['net_convert'] == 2 and calls['host_mc2sp'] == 1                      # the literal path: two uploads, host mc2sp
    assert float(numpy.abs(out.sp / plain.sp - 1).max()) < 2e-5 and numpy.array_equal(out.mc, plain.mc)
    assert numpy.array_equal(out.f0, plain.f0) and numpy.array_equal(out.ap, plain.ap)
    monkeypatch.delenv('RY_FUSE_STEPS')
    # the lazy spectrogram is indistinguishable from the array for anybody else
    f_eff, eff = ac.separate_effective(wave=wave, feature=vc_mod.AcousticFeatureWrapper(wave=wave, **feat), threshold=60)
    f = ac.decode_spectrogram(ac.combine_silent(effective=eff, feature=ac.convert(f_eff)))
    assert isinstance(f.sp, fusion.LazySpectrogram) and f.sp.shape == (N, 513) and len(f.sp) == N and f.sp.dtype == numpy.float64
    host = real_fast(f.mc, sptk.mcepalpha(FS), 1024)
    f.sp += 1e-16
    lazy32 = f.sp.astype(numpy.float32)
    assert isinstance(lazy32, fusion.LazySpectrogram) and lazy32.dtype == numpy.float32
    arr = numpy.asarray(lazy32)                                                        # a numpy function materialises it (ry_vc_mid_sp)
    assert arr.dtype == numpy.float32 and float(numpy.abs(arr / (host + 1e-16) - 1).max()) < 2e-5 and calls['mid_sp'] == 1
    assert float(numpy.abs(f.sp[3:5] / (host[3:5] + 1e-16) - 1).max()) < 2e-5         # indexing too
    # a second SuperResolution shim is not the one the object was built for: it gets the array (and the right answer)
    _, sr2 = build_converters(models)
    n_before = calls['stage2_from_mc']
    y2 = sr2.convert(f.sp.astype(numpy.float32))
    assert calls['stage2_from_mc'] == n_before and float(numpy.abs(y2 / plain.sp - 1).max()) < 2e-5
    # after another convert() the device rows are gone: the stale object falls back to the host formula
    g = ac.decode_spectrogram(ac.combine_silent(effective=eff, feature=ac.convert(f_eff)))
    ac.convert(f_eff)
    stale = numpy.asarray(g.sp)
    assert float(numpy.abs(stale / host - 1).max()) < 1e-12 and stale.dtype == numpy.float64
    # RY_SR_DISCARD (opt-in, for run.py unchanged): the frames ConvertStream.process picks away are not computed -- through the fused
    # continuation (ry_vc_stage2_from_mc) and through the plain array call of the shim; the kept rows do not move by a bit
    monkeypatch.setenv('RY_SR_DISCARD', '7,5')
    part = run()
    assert numpy.array_equal(part.sp[7:N - 5], out.sp[7:N - 5]) and not part.sp[:7].any() and not part.sp[N - 5:].any()
    assert numpy.array_equal(part.mc, out.mc)
    arr_part = sr.convert(numpy.asarray(plain.sp, dtype=numpy.float32))
    monkeypatch.delenv('RY_SR_DISCARD')
    arr_full = sr.convert(numpy.asarray(plain.sp, dtype=numpy.float32))
    assert numpy.array_equal(arr_part[7:N - 5], arr_full[7:N - 5]) and not arr_part[:7].any() and arr_full[:7].any()
    again = run()
    assert numpy.array_equal(again.sp, out.sp)                                        # ... and without the variable every frame is back
    for m in [k for k in sys.modules if k.startswith('realtime_voice_conversion')]:
        sys.modules.pop(m)


@pytest.mark.skipif(not REF.exists(), reason='/root/reference is not mounted here')
def test_convert_worker_mirror_matches_the_reference_loop_and_stays_bounded(models, on_emulator, monkeypatch):
    """`realtime_yukarin_amd.worker.convert_worker` over FeatureQueues against the reference's loop body
    (convert_worker.py:33-57) run inline: same windows out, while the mirror's stream no longer grows."""
    import threading
    for p in (str(ROOT / 'tests' / 'stubs'), str(REF)):
        monkeypatch.syspath_prepend(p)
    vc_mod = importlib.import_module('realtime_voice_conversion.yukarin_wrapper.voice_changer')
    st_mod = importlib.import_module('realtime_voice_conversion.stream')
    util = importlib.import_module('realtime_voice_conversion.worker.utility')
    from realtime_yukarin_amd import worker
    from realtime_yukarin_amd.transport import FeatureQueue
    ac, sr = build_converters(models)
    time_length, extra_time, n_items = 0.3, 0.1, 5
    rng = numpy.random.default_rng(5)
    inputs = []
    for _ in range(n_items):
        wave, feat = make_input(rng)
        inputs.append(vc_mod.AcousticFeatureWrapper(wave=wave, **feat))

    # the reference's loop body, inline
    stream = st_mod.ConvertStream(voice_changer=vc_mod.VoiceChanger(super_resolution=sr, acoustic_converter=ac, threshold=60))
    wrapper = st_mod.StreamWrapper(stream=stream, extra_time=extra_time)
    want, start_time = [], extra_time
    for f in inputs:
        stream.add(start_time=start_time, data=f)
        start_time += time_length
        want.append(wrapper.process_next(time_length=time_length))
    assert len(stream.stream) == n_items                                      # never trimmed

    q_in, q_out = FeatureQueue(slots=4, slot_bytes=4 << 20), FeatureQueue(slots=4, slot_bytes=4 << 20)
    lock = threading.Lock(); lock.acquire()
    seen = {}
    real_remove = st_mod.ConvertStream.remove

    def counting_remove(self, end_time):
        real_remove(self, end_time)
        seen['segments'] = max(seen.get('segments', 0), len(self.stream))
    monkeypatch.setattr(st_mod.ConvertStream, 'remove', counting_remove)
    t = threading.Thread(target=worker.convert_worker, args=(ac, sr, time_length, extra_time, 60, q_in, q_out, lock), daemon=True)
    t.start()
    for i, f in enumerate(inputs):
        q_in.put(util.Item(item=f, index=i))
        got = q_out.get(timeout=300)
        assert got.index == i and got.item.sp.shape == want[i].sp.shape == (60, 513)
        # the reference's VoiceChanger goes step by step (host mc2sp in float64), the mirror's fused core does mc2sp in fp32
        assert float(numpy.abs(got.item.sp / want[i].sp - 1).max()) < 2e-5 and numpy.array_equal(got.item.f0, want[i].f0)
        assert numpy.array_equal(got.item.ap, want[i].ap) and numpy.array_equal(got.item.voiced, want[i].voiced)   # (mc is not a key of the output segments, feature_segment.py:20)
    q_in.put(None)
    t.join(timeout=30)
    assert not t.is_alive() and not lock.locked()
    assert seen['segments'] <= 3                                              # bounded by the overlap, not by the run length
    q_in.close(); q_out.close()
    # a backlog of items: the worker keeps two windows in flight (VoiceChanger.begin / finish) and still returns the same windows in order
    from realtime_yukarin_amd.voice_changer import VoiceChanger as MirrorVC
    flight = {'now': 0, 'max': 0}
    real_begin, real_finish = MirrorVC.begin, MirrorVC.finish

    def begin(self, f, discard=(0, 0)):
        flight['now'] += 1; flight['max'] = max(flight['max'], flight['now'])
        flight['discard'] = discard
        return real_begin(self, f, discard)

    def finish(self, h):
        flight['now'] -= 1
        return real_finish(self, h)
    monkeypatch.setattr(MirrorVC, 'begin', begin); monkeypatch.setattr(MirrorVC, 'finish', finish)
    q_in, q_out = FeatureQueue(slots=8, slot_bytes=4 << 20), FeatureQueue(slots=8, slot_bytes=4 << 20)
    for i, f in enumerate(inputs):
        q_in.put(util.Item(item=f, index=i))
    q_in.put(None)
    lock2 = threading.Lock(); lock2.acquire()
    t2 = threading.Thread(target=worker.convert_worker, args=(ac, sr, time_length, extra_time, 60, q_in, q_out, lock2), daemon=True)
    t2.start()
    for i in range(n_items):
        got = q_out.get(timeout=300)
        assert got.index == i and float(numpy.abs(got.item.sp / want[i].sp - 1).max()) < 2e-5 and numpy.array_equal(got.item.f0, want[i].f0)
    t2.join(timeout=30)
    assert not t2.is_alive() and flight['max'] == 2 and flight['now'] == 0
    assert flight['discard'] == (20, 20)                                      # the worker announces the frames `pick` drops: stage 2 does not compute them
    q_in.close(); q_out.close()
    for m in [k for k in sys.modules if k.startswith('realtime_voice_conversion')]:
        sys.modules.pop(m)


def test_converters_may_be_closed_or_rebuilt_under_the_voice_changer(models, on_emulator):
    """A converter shim that is closed (or whose predictor is rebuilt for another bin count) while a VoiceChanger still holds the window core
    built on it: the next window rebuilds the core on the new predictors instead of submitting on freed ones, and `ac.close(); sr.close();
    vc.close()` in that order is safe (the round-2 advisor's use-after-free)."""
    from realtime_yukarin_amd.voice_changer import VoiceChanger
    from yukarin import AcousticFeature
    ac, sr = build_converters(models)
    wave, feat = make_input(numpy.random.default_rng(23))

    def f_in():
        f = AcousticFeature(**{k: v.copy() for k, v in feat.items()}); f.wave = wave
        return f
    vc = VoiceChanger(acoustic_converter=ac, super_resolution=sr, threshold=60)
    first = vc.convert_from_acoustic_feature(f_in())
    core = vc._fused_core()
    sr.convert(numpy.full((10, 129), 1e-3, numpy.float32))             # another bin count: the shim frees its predictor and builds a new one
    assert core.handle is None, 'the core on the freed predictor went with it'
    again = vc.convert_from_acoustic_feature(f_in())                    # rebuilt on the 513-bin predictor
    assert vc._fused_core() is not core and numpy.array_equal(again.sp, first.sp) and numpy.array_equal(again.mc, first.mc)
    ac.close()
    third = vc.convert_from_acoustic_feature(f_in())
    assert numpy.array_equal(third.sp, first.sp)
    ac.close(); sr.close(); vc.close(); vc.close()
    check(first, expected(models, ac, wave, feat), feat)
