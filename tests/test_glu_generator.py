"""`model.glu_generator: true` stage-1 configs (BASELINE.json north_star: "1-D/2-D dilated conv + GLU"; loaded by whatever config.json the
converter at /root/reference/realtime_voice_conversion/converter/yukarin_converter.py:39-47 is pointed at).  The gated predictor class is
in the un-vendored `yukarin` package, so the graph built here is THIS repository's reading of it -- UNVERIFIED [MEM]: every conv +
BatchNormalization block computes twice the channels and is gated, h[:C] * sigmoid(h[C:]) -- and loads only behind
RY_ALLOW_UNVERIFIED_GLU=1; the strict K-list / shape validation decides whether a real model fits.  Parity: the HIP path (the generic
weight-streaming kernels with the gate applied in the consumer's staging) against two restatements (numpy, torch), emulator and GPU."""
import json

import numpy
import pytest

from conftest import rel_max
from oracle import torch_ref, unet
from realtime_yukarin_amd import compat, engine, synth
from realtime_yukarin_amd.netspec import NetDesc, param_count, param_list
from realtime_yukarin_amd.weights import flatten_params, save_npz, synthetic_params

compat.install()
TOL = 1e-4


def check_net(ctx, base, n_frames, e=8, in_ch=9):
    d = NetDesc(1, in_ch, 9, base, e, glu=True)
    P = synthetic_params(d, 470 + base, bias_std=0.05)
    shapes = dict(param_list(d))
    assert shapes['encoder/c1/c/W'][0] == 2 * 2 * base and shapes['encoder/c1/batchnorm/gamma'] == (2 * 2 * base,)   # value | gate
    assert shapes['encoder/c2/c/W'][1] == 2 * base                                                                # ... consumers see half
    net = engine.Net(ctx, d, flatten_params(d, P))
    x = synth.stage1_input(n_frames, seed=480 + base)[0][:, :in_ch] if in_ch <= 9 else synth.stage1_input(n_frames, stress=True)[0]
    y = net.convert(x)
    r_t = torch_ref.stage1_convert_core(torch_ref.TorchUNet(P, e, glu=True), x)
    assert y.shape == r_t.shape == (n_frames, 9)
    e_t = rel_max(y, r_t)
    if base <= 16:                                                     # the numpy restatement too where it is quick
        assert rel_max(unet.stage1_convert_core(x, P, e, glu=True), r_t) < 1e-5
    assert numpy.array_equal(y, net.convert(x)), 'graph replay must be deterministic'
    plain = NetDesc(1, in_ch, 9, base, e)
    assert param_count(d) > param_count(plain)
    net.close()
    return e_t


def test_glu_predictor_emu(emu_ctx):
    assert check_net(emu_ctx, 8, 60) < TOL
    assert check_net(emu_ctx, 16, 37, e=5) < TOL                       # 'same' 1x1 blocks gated as well


@pytest.mark.gpu
def test_glu_predictor_gpu(gpu_ctx):
    for base, n in ((8, 100), (64, 300), (64, 128)):
        e = check_net(gpu_ctx, base, n)
        print('glu_generator stage-1, base %d, %d frames: max-norm error vs the torch restatement %.2e' % (base, n, e))
        assert e < TOL


def test_glu_config_loads_only_behind_the_flag(tmp_path, emu_ctx, monkeypatch):
    from yukarin import AcousticConverter, AcousticFeature
    from yukarin.config import create_from_json
    monkeypatch.setattr(engine, 'get_context', lambda device=0, lib=None: emu_ctx)
    d = NetDesc(1, 9, 9, 8, 8, glu=True)
    P = synthetic_params(d, 471)
    save_npz(tmp_path / 'g.npz', P)
    (tmp_path / 'g.json').write_text(json.dumps({
        'dataset': {'acoustic_param': {'sampling_rate': 16000, 'frame_period': 5, 'order': 8, 'alpha': 0.41}, 'in_features': ['mc'], 'out_features': ['mc']},
        'model': {'in_channels': 9, 'out_channels': 9, 'generator_base_channels': 8, 'generator_extensive_layers': 8, 'glu_generator': True}}))
    monkeypatch.delenv('RY_ALLOW_UNVERIFIED_GLU', raising=False)
    with pytest.raises(NotImplementedError, match='UNVERIFIED'):
        create_from_json(tmp_path / 'g.json')
    monkeypatch.setenv('RY_ALLOW_UNVERIFIED_GLU', '1')
    cfg = create_from_json(tmp_path / 'g.json')
    ac = AcousticConverter(cfg, tmp_path / 'g.npz', gpu=0)
    assert ac.desc.glu
    x = synth.stage1_input(50, seed=3)[0]
    out = ac.convert(AcousticFeature(mc=x, f0=numpy.zeros((50, 1), numpy.float32), ap=numpy.zeros((50, 513), numpy.float32), voiced=numpy.zeros((50, 1), bool)))
    assert rel_max(out.mc, torch_ref.stage1_convert_core(torch_ref.TorchUNet(P, 8, glu=True), x)) < TOL
    # a plain (ungated) weight file does not fit the gated K-list: refused, never mis-mapped
    save_npz(tmp_path / 'plain.npz', synthetic_params(NetDesc(1, 9, 9, 8, 8), 472))
    with pytest.raises(ValueError, match='shape'):
        AcousticConverter(cfg, tmp_path / 'plain.npz', gpu=0)
    ac.close()
