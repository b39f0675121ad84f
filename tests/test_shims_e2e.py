"""The drop-in layer end to end: model files on disk (Chainer save_npz key layout + config.json) -> `yukarin.AcousticConverter` /
`become_yukarin.SuperResolution` shims -> pickle round trip (run.py ships them to a child Process) -> the mirror
`VoiceChanger.convert_from_acoustic_feature` (/root/reference/realtime_voice_conversion/yukarin_wrapper/voice_changer.py:24-42) in
its fused and step-by-step forms and `convert_windows`, against the COMPOSED ORACLE: the independent loop-per-frame silence gate
(oracle/effective_frame.py), the torch/oneDNN CNNs (oracle/torch_ref.py) and the independent restatement of pysptk.mc2sp (oracle/mc2sp.py), element
by element.  The same body runs on the real GPU with SYN-64 at the BASELINE config #3 window (-m gpu) and on the emulator with SYN-8
(CPU suite), plus `SuperResolution.convert` alone at the config #1 / #2 windows and for a batch of 8 windows."""
import json
import pickle

import numpy
import pytest

from oracle import effective_frame as oef
from oracle import mc2sp as omc
from oracle import torch_ref
from realtime_yukarin_amd import compat, engine, synth
from realtime_yukarin_amd.weights import save_npz

compat.install()
FS, FRAME_PERIOD, HOP = 16000, 5, 80
TOL = 1e-4                      # north star: 1e-4 relative fp32


def rel_elem(y, ref, floor):
    """max over elements of |y - ref| / max(|ref|, floor): element-wise relative error with an absolute floor (mel-cepstrum
    coefficients cross zero; `floor` = 1e-2 of the coefficient's scale keeps the ratio meaningful there)."""
    y = numpy.asarray(y, numpy.float64); ref = numpy.asarray(ref, numpy.float64)
    return float((numpy.abs(y - ref) / numpy.maximum(numpy.abs(ref), floor)).max())


def write_models(d, name, out_rate=FS):
    (d1, P1), (d2, P2) = synth.model_params(name)
    save_npz(d / 's1.npz', P1)
    save_npz(d / 's2.npz', P2)
    (d / 's1.json').write_text(json.dumps({
        'dataset': {'acoustic_param': {'sampling_rate': FS, 'frame_period': FRAME_PERIOD, 'order': 8, 'alpha': 0.41},
                    'in_features': ['mc'], 'out_features': ['mc']},
        'model': {'in_channels': 9, 'out_channels': 9, 'generator_base_channels': d1.base, 'generator_extensive_layers': 8}}))
    (d / 's2.json').write_text(json.dumps({
        'dataset': {'param': {'voice_param': {'sample_rate': out_rate}, 'acoustic_feature_param': {'frame_period': FRAME_PERIOD, 'order': 8}}},
        'model': {'generator_base_channels': d2.base, 'generator_extensive_layers': 8}}))
    numpy.save(str(d / 'in_stat.npy'), {'mean': numpy.log(200.0), 'var': 0.04})
    numpy.save(str(d / 'tg_stat.npy'), {'mean': numpy.log(300.0), 'var': 0.09})
    return P1, P2


def build_converters(d, out_rate=FS):
    from become_yukarin import SuperResolution
    from become_yukarin.config.sr_config import create_from_json as create_sr_config
    from yukarin import AcousticConverter
    from yukarin.config import create_from_json as create_config
    from yukarin.f0_converter import F0Converter
    f0c = F0Converter(input_statistics=d / 'in_stat.npy', target_statistics=d / 'tg_stat.npy')
    ac = AcousticConverter(create_config(d / 's1.json'), d / 's1.npz', gpu=0, f0_converter=f0c, out_sampling_rate=out_rate)
    sr = SuperResolution(create_sr_config(d / 's2.json'), d / 's2.npz', gpu=0)
    return pickle.loads(pickle.dumps(ac)), pickle.loads(pickle.dumps(sr))          # what crosses the Process boundary in run.py:69-79


def vc_alpha(ac):
    ac.mc2sp_matrix()
    return round(ac._alpha_out, 3)


def make_window(n, seed):
    """A window with a loud part, a silent stretch and a quiet (below the gate) stretch."""
    rng = numpy.random.default_rng(seed)
    wave = (0.1 * rng.normal(size=n * HOP)).astype(numpy.float32)
    a, b = n // 6, n // 2
    wave[a * HOP:b * HOP] = 0.0
    wave[(n - n // 8) * HOP:] *= 1e-5
    f0 = numpy.where(rng.random((n, 1)) < 0.3, 0.0, rng.lognormal(numpy.log(220.0), 0.2, (n, 1))).astype(numpy.float32)
    feat = dict(f0=f0, ap=rng.uniform(0.001, 0.999, (n, 513)).astype(numpy.float32),
                mc=(rng.normal(size=(n, 9)) * synth.MC_SCALE).astype(numpy.float32), voiced=f0 > 0)
    return wave, feat


def expected(t1, t2, f0c, wave, feat, n, threshold, out_rate=FS):
    """voice_changer.py:24-42 step by step on the oracle."""
    eff = oef.separate_effective_mask(wave, FS, n, threshold, 1024, FRAME_PERIOD)
    mc = numpy.zeros((n, 9), numpy.float32)
    if eff.any():
        mc[eff] = torch_ref.stage1_convert_core(t1, feat['mc'][eff])
    f0 = numpy.zeros((n, 1), numpy.float32)
    lf = numpy.log(numpy.where(feat['f0'] > 0, feat['f0'], 1.0).astype(numpy.float64))
    conv = numpy.exp((numpy.sqrt(0.09) / numpy.sqrt(0.04)) * (lf - numpy.log(200.0)) + numpy.log(300.0))
    f0[eff] = numpy.where(feat['f0'] > 0, conv, 0.0).astype(numpy.float32)[eff]
    ap = numpy.zeros((n, 513), numpy.float32); ap[eff] = feat['ap'][eff]
    sp_mid = (omc.mc2sp(mc, omc.mcepalpha(out_rate), 1024) + 1e-16).astype(numpy.float32)
    return dict(eff=eff, mc=mc, f0=f0, ap=ap, sp=torch_ref.stage2_convert(t2, sp_mid))


def check(out, exp, n, tag, sp_tol=TOL):
    assert out.sp.shape == (n, 513) and out.sp.dtype == numpy.float32 and numpy.isfinite(out.sp).all()
    e_sp = float(numpy.abs(out.sp.astype(numpy.float64) / exp['sp'] - 1).max())                   # element-wise
    e_mc_max = float(numpy.abs(out.mc - exp['mc']).max() / numpy.abs(exp['mc']).max())          # max-norm (round-1 metric)
    e_mc_el = max(rel_elem(out.mc[:, c], exp['mc'][:, c], 1e-2 * synth.MC_SCALE[c]) for c in range(9))   # element-wise, floored per coefficient
    print('%s: sp element-wise %.2e, mc max-norm %.2e, mc element-wise (floor 1e-2 of the coefficient scale) %.2e' % (tag, e_sp, e_mc_max, e_mc_el))
    assert e_sp < sp_tol and e_mc_max < TOL and e_mc_el < TOL
    assert not out.mc[~exp['eff']].any()
    assert numpy.allclose(out.f0, exp['f0'], rtol=1e-6) and numpy.array_equal(out.ap, exp['ap'])
    assert 0 < exp['eff'].sum() < n


def run_voice_changer_e2e(d, name, n, dtype_env=None, sp_tol=TOL, out_rate=FS):
    from realtime_yukarin_amd.voice_changer import VoiceChanger
    from yukarin import AcousticFeature, Wave
    P1, P2 = write_models(d, name, out_rate)
    t1, t2 = torch_ref.TorchUNet(P1), torch_ref.TorchUNet(P2)
    ac, sr = build_converters(d, out_rate)
    assert vc_alpha(ac) == {16000: 0.41, 24000: 0.466}[out_rate]

    class Wrapped(AcousticFeature):                                              # AcousticFeatureWrapper equivalent
        pass

    def f_in(wave, feat):
        f = Wrapped(**{k: v.copy() for k, v in feat.items()}); f.wave = Wave(wave=wave, sampling_rate=FS)
        return f
    wave, feat = make_window(n, 21)
    exp = expected(t1, t2, ac.f0_converter, wave, feat, n, 60, out_rate)
    vc = VoiceChanger(acoustic_converter=ac, super_resolution=sr, threshold=60)
    assert vc._fused_core() is not None
    out = vc.convert_from_acoustic_feature(f_in(wave, feat))
    check(out, exp, n, '%s fused n=%d%s' % (name, n, ' ' + dtype_env if dtype_env else ''), sp_tol)
    vc2 = VoiceChanger(acoustic_converter=ac, super_resolution=sr, threshold=60)
    vc2._fused_core = lambda: None
    gen = vc2.convert_from_acoustic_feature(f_in(wave, feat))
    check(gen, exp, n, '%s step-by-step n=%d' % (name, n), sp_tol)
    # convert_windows: three different windows, stage-2 in one batched call, each against its own oracle
    wins = [make_window(n, 30 + i) for i in range(3)]
    outs = vc.convert_windows([f_in(w, f) for w, f in wins])
    for i, (w, f) in enumerate(wins):
        check(outs[i], expected(t1, t2, ac.f0_converter, w, f, n, 60, out_rate), n, '%s convert_windows[%d]' % (name, i), sp_tol)
    return sr, t2


# ---------------------------------------------------------------- CPU suite: the same body on the emulator, SYN-8
@pytest.fixture()
def on_emulator(emu_ctx, monkeypatch):
    monkeypatch.setattr(engine, 'get_context', lambda device=0, lib=None: emu_ctx)
    return emu_ctx


def test_shims_end_to_end_emu(tmp_path, on_emulator):
    run_voice_changer_e2e(tmp_path, 'SYN-8', 60)


def test_shims_end_to_end_24khz_emu(tmp_path, on_emulator):
    """out_sampling_rate = 24000 -> alpha 0.466: the rate converter/yukarin_converter.py:46 hard-codes for run.py."""
    run_voice_changer_e2e(tmp_path, 'SYN-8', 60, out_rate=24000)


# ---------------------------------------------------------------- GPU suite: SYN-64 at BASELINE's windows
@pytest.mark.gpu
def test_shims_end_to_end_gpu(tmp_path, gpu_ctx):
    """BASELINE config #3 window (0.5 s buffer + 2 x 0.5 s extra = 300 frames), exact fp32."""
    run_voice_changer_e2e(tmp_path, 'SYN-64', 300)


@pytest.mark.gpu
@pytest.mark.parametrize('n_frames', [100, 400, 600])      # config #3 / #4 core, config #5 (1.0 s + 2 x 0.5 s), config #1 (check.py)
def test_shims_end_to_end_gpu_other_windows(tmp_path, gpu_ctx, n_frames):
    """The chained core (fused, step by step through the lazy spectrogram, convert_windows) at the other BASELINE windows."""
    run_voice_changer_e2e(tmp_path, 'SYN-64', n_frames)


@pytest.mark.gpu
def test_shims_end_to_end_gpu_24khz(tmp_path, gpu_ctx):
    """The whole path with out_sampling_rate = 24000 (alpha = 0.466), the value /root/reference/realtime_voice_conversion/converter/
    yukarin_converter.py:46 hard-codes for run.py: the fused core's mc2sp matrix is built for that rate."""
    run_voice_changer_e2e(tmp_path, 'SYN-64', 300, out_rate=24000)


@pytest.mark.gpu
def test_shims_end_to_end_gpu_split_bf16(tmp_path, gpu_ctx, monkeypatch):
    """The same through RY_SR_DTYPE=bf16x3 (how an unchanged run.py opts into the split-bf16 stage-2 arithmetic): same 1e-4 bar."""
    monkeypatch.setenv('RY_SR_DTYPE', 'bf16x3')
    run_voice_changer_e2e(tmp_path, 'SYN-64', 300, dtype_env='bf16x3')


@pytest.mark.gpu
@pytest.mark.parametrize('n_frames', [600, 1000])                    # BASELINE config #1 (check.py: 600) and #2 (1000)
def test_super_resolution_shim_at_the_large_windows_gpu(tmp_path, gpu_ctx, n_frames):
    P1, P2 = write_models(tmp_path, 'SYN-64')
    _, sr = build_converters(tmp_path)
    t2 = torch_ref.TorchUNet(P2)
    sp = synth.stage2_input(n_frames)[0]
    y = sr.convert(sp)
    r = torch_ref.stage2_convert(t2, sp)
    e = float(numpy.abs(y.astype(numpy.float64) / r - 1).max())
    print('SuperResolution.convert n=%d: element-wise %.2e' % (n_frames, e))
    assert y.shape == (n_frames, 513) and e < TOL
    assert numpy.array_equal(y[:, -1], y[:, -2])


@pytest.mark.gpu
def test_super_resolution_batch_of_8_windows_against_the_oracle_gpu(tmp_path, gpu_ctx):
    """Eight different 300-frame windows in ONE `ry_sr_convert` call (chunk parallelism inside a GPU), each against the oracle."""
    P1, P2 = write_models(tmp_path, 'SYN-64')
    _, sr = build_converters(tmp_path)
    t2 = torch_ref.TorchUNet(P2)
    sp = synth.stage2_input(300, windows=8, seed=901)
    y = sr._get_net(513).convert(sp)
    worst = 0.0
    for w in range(8):
        worst = max(worst, float(numpy.abs(y[w].astype(numpy.float64) / torch_ref.stage2_convert(t2, sp[w]) - 1).max()))
    print('batch of 8 windows: worst element-wise %.2e' % worst)
    assert worst < TOL


@pytest.mark.gpu
def test_stage1_elementwise_next_to_max_norm_gpu(gpu_ctx):
    """Stage-1 parity in both metrics at the BASELINE windows: max-norm (max|d| / max|ref|) and element-wise with a floor of
    1e-2 of each coefficient's scale (the mel-cepstrum spans 20x in scale across its 9 coefficients)."""
    from realtime_yukarin_amd.weights import flatten_params
    (d1, P1), _ = synth.model_params('SYN-64')
    n1 = engine.Net(gpu_ctx, d1, flatten_params(d1, P1))
    t1 = torch_ref.TorchUNet(P1)
    for n in (100, 300, 600, 1000):
        x = synth.stage1_input(n)[0]
        y, r = n1.convert(x), torch_ref.stage1_convert_core(t1, x)
        e_max = float(numpy.abs(y - r).max() / numpy.abs(r).max())
        e_el = max(rel_elem(y[:, c], r[:, c], 1e-2 * float(numpy.abs(r[:, c]).max())) for c in range(9))
        print('stage-1 n=%d: max-norm %.2e, element-wise (floor 1e-2 of the column max) %.2e' % (n, e_max, e_el))
        assert e_max < TOL and e_el < TOL
    n1.close()


@pytest.mark.gpu
def test_shims_end_to_end_gpu_config5_bf16_at_400_frames(tmp_path, gpu_ctx, monkeypatch):
    """BASELINE config #5 through the import surface: `RY_SR_DTYPE=bf16` (how an unchanged run.py opts in), buffer_time 1.0 s + 2 x 0.5 s =
    400 frames, the mirror VoiceChanger's fused call.  mc / f0 / ap as exact as ever (stage 1 is fp32), the spectrogram inside the stated
    3e-2 on the log-spectrum against the fp32 oracle."""
    from realtime_yukarin_amd.voice_changer import VoiceChanger
    from yukarin import AcousticFeature, Wave
    monkeypatch.setenv('RY_SR_DTYPE', 'bf16')
    n = 400
    P1, P2 = write_models(tmp_path, 'SYN-64')
    t1, t2 = torch_ref.TorchUNet(P1), torch_ref.TorchUNet(P2)
    ac, sr = build_converters(tmp_path)
    wave, feat = make_window(n, 55)
    f = AcousticFeature(**{k: v.copy() for k, v in feat.items()}); f.wave = Wave(wave=wave, sampling_rate=FS)
    exp = expected(t1, t2, ac.f0_converter, wave, feat, n, 60)
    vc = VoiceChanger(acoustic_converter=ac, super_resolution=sr, threshold=60)
    assert vc._fused_core() is not None
    out = vc.convert_from_acoustic_feature(f)
    e_log = float(numpy.abs(numpy.log(out.sp.astype(numpy.float64)) - numpy.log(exp['sp'])).max() / numpy.abs(numpy.log(exp['sp'])).max())
    e_mc = float(numpy.abs(out.mc - exp['mc']).max() / numpy.abs(exp['mc']).max())
    print('config #5 through the shims, 400 frames, bf16 stage 2: log-spectrum %.2e (stated 3e-2), mc %.2e' % (e_log, e_mc))
    assert 1e-5 < e_log < 3e-2 and e_mc < TOL and not out.mc[~exp['eff']].any()
    assert numpy.allclose(out.f0, exp['f0'], rtol=1e-6) and numpy.array_equal(out.ap, exp['ap'])
    vc.close(); ac.close(); sr.close()


# ---------------------------------------------------------------- a NON-canonical stage-1 configuration: "mel + f0" in, mel out
def _noncanonical_models(d, in_features):
    """Model files whose stage 1 reads more than the mel-cepstrum (north star: "(mel/f0/ap) frame blocks"): in_channels = the summed widths."""
    widths = {'mc': 9, 'f0': 1, 'ap': 513}
    cin = sum(widths[k] for k in in_features)
    (d1, P1), (d2, P2) = synth.model_params('SYN-8', stage1_in=cin)
    save_npz(d / 's1.npz', P1)
    save_npz(d / 's2.npz', P2)
    (d / 's1.json').write_text(json.dumps({
        'dataset': {'acoustic_param': {'sampling_rate': FS, 'frame_period': FRAME_PERIOD, 'order': 8, 'alpha': 0.41},
                    'in_features': list(in_features), 'out_features': ['mc']},
        'model': {'in_channels': cin, 'out_channels': 9, 'generator_base_channels': d1.base, 'generator_extensive_layers': 8}}))
    (d / 's2.json').write_text(json.dumps({
        'dataset': {'param': {'voice_param': {'sample_rate': FS}, 'acoustic_feature_param': {'frame_period': FRAME_PERIOD, 'order': 8}}},
        'model': {'generator_base_channels': d2.base, 'generator_extensive_layers': 8}}))
    numpy.save(str(d / 'in_stat.npy'), {'mean': numpy.log(200.0), 'var': 0.04})
    numpy.save(str(d / 'tg_stat.npy'), {'mean': numpy.log(300.0), 'var': 0.09})
    return P1, P2


def test_non_canonical_in_features_take_the_generic_path_and_whole_objects_emu(tmp_path, on_emulator):
    """`in_features = ['mc', 'f0']` (C_in = 10): `AcousticConverter.convert` concatenates the columns (`encode_feature`), the window is NOT
    fusable, so the mirror VoiceChanger runs the reference's step order call by call (stage-1 on the GPU, mc2sp on the host, stage-2 on
    the GPU) -- against the composed oracle; and the dispatcher notices by itself that such windows cannot travel without their `ap`
    block / as bare arrays (`lean` off) and returns the same bits as the in-process call."""
    from realtime_yukarin_amd import dispatch
    from realtime_yukarin_amd.voice_changer import VoiceChanger
    from yukarin import AcousticFeature, Wave
    from dispatch_hooks import emu_hook
    n = 64
    P1, P2 = _noncanonical_models(tmp_path, ('mc', 'f0'))
    t1, t2 = torch_ref.TorchUNet(P1), torch_ref.TorchUNet(P2)
    ac, sr = build_converters(tmp_path)
    assert not ac.fusable() and ac.desc.in_ch == 10
    wave, feat = make_window(n, 77)

    def f_in():
        f = AcousticFeature(**{k: v.copy() for k, v in feat.items()}); f.wave = Wave(wave=wave, sampling_rate=FS)
        return f
    vc = VoiceChanger(acoustic_converter=ac, super_resolution=sr, threshold=60)
    assert vc._fused_core() is None
    out = vc.convert_from_acoustic_feature(f_in())
    # the composed oracle with the concatenated input columns
    eff = oef.separate_effective_mask(wave, FS, n, 60, 1024, FRAME_PERIOD)
    assert 0 < eff.sum() < n
    x = numpy.concatenate([feat['mc'], feat['f0']], axis=1)[eff]
    mc = numpy.zeros((n, 9), numpy.float32); mc[eff] = torch_ref.stage1_convert_core(t1, x)
    sp_mid = (omc.mc2sp(mc, omc.mcepalpha(FS), 1024) + 1e-16).astype(numpy.float32)
    sp = torch_ref.stage2_convert(t2, sp_mid)
    assert float(numpy.abs(out.mc - mc).max() / numpy.abs(mc).max()) < TOL and not out.mc[~eff].any()
    assert float(numpy.abs(out.sp.astype(numpy.float64) / sp - 1).max()) < TOL
    assert numpy.array_equal(out.ap[eff], feat['ap'][eff]) and not out.ap[~eff].any()
    with dispatch.ChunkDispatcher(ac, sr, [0], threshold=60, comm='host', worker_hook=emu_hook, start_timeout=900) as d:
        assert d.lean is False
        d.submit(0, f_in(), discard=(7, 7), pick=(7, -7, ('f0', 'ap', 'sp', 'voiced', 'mc')))
        d.submit(1, f_in())
        (_, a), (_, b) = d.drain(timeout=900)
    for k in ('f0', 'ap', 'sp', 'voiced', 'mc'):
        assert numpy.array_equal(getattr(b, k), getattr(out, k)), k
        assert numpy.array_equal(getattr(a, k), getattr(out, k)[7:-7]), k
    vc.close(); ac.close(); sr.close()
