"""The CPU oracle pinned: hand-written known-answer cases, two independent implementations agreeing, and the
committed golden fixtures (PARITY UNPINNED at the reference's own boundary -- see oracle/__init__.py)."""
import glob
from pathlib import Path

import numpy
import pytest

from oracle import ops_numpy as ops
from oracle import torch_ref, unet
from realtime_yukarin_amd.netspec import NetDesc
from realtime_yukarin_amd.weights import synthetic_params

GOLD = Path(__file__).resolve().parent / 'golden'


def test_conv1d_delta_returns_the_taps():
    """Cross-correlation: a unit impulse at position p gives y[p + pad - k] = W[k] (taps appear reversed in y)."""
    W = numpy.arange(1, 5, dtype='f4').reshape(1, 1, 4)                  # taps 1,2,3,4
    x = numpy.zeros((1, 1, 8), 'f4'); x[0, 0, 4] = 1.0
    y = ops.conv_nd(x, W, None, stride=1, pad=1)                          # y[l] = sum_k W[k] x[l - 1 + k]
    assert y.shape == (1, 1, 7)
    assert numpy.array_equal(y[0, 0], [0, 0, 4, 3, 2, 1, 0])


def test_conv1d_k4s2p1_known_answer():
    W = numpy.array([1., 10., 100., 1000.], 'f4').reshape(1, 1, 4)
    x = numpy.arange(1, 7, dtype='f4').reshape(1, 1, 6)                   # 1..6, zero padded
    y = ops.conv_nd(x, W, None, stride=2, pad=1)                          # y[l] = sum_k W[k] x[2l - 1 + k]
    assert numpy.array_equal(y[0, 0], [0 + 10 + 200 + 3000, 2 + 30 + 400 + 5000, 4 + 50 + 600 + 0])


def test_deconv1d_k4s2p1_of_a_one_hot():
    """Transposed conv scatters: out[2i - 1 + k] += W[k] x[i]; length doubles."""
    W = numpy.array([1., 2., 3., 4.], 'f4').reshape(1, 1, 4)
    x = numpy.zeros((1, 1, 3), 'f4'); x[0, 0, 1] = 1.0
    y = ops.deconv_nd(x, W, None, stride=2, pad=1)
    assert y.shape == (1, 1, 6)
    assert numpy.array_equal(y[0, 0], [0, 1, 2, 3, 4, 0])


def test_deconv2d_weight_layout_is_cin_cout():
    W = numpy.zeros((2, 3, 4, 4), 'f4'); W[1, 2, 0, 3] = 5.0              # (Cin, Cout, ky, kx)
    x = numpy.zeros((1, 2, 2, 2), 'f4'); x[0, 1, 1, 0] = 2.0
    y = ops.deconv_nd(x, W, None, stride=2, pad=1)                        # out[2*1 - 1 + 0][2*0 - 1 + 3] = 10
    assert y.shape == (1, 3, 4, 4)
    want = numpy.zeros_like(y); want[0, 2, 1, 2] = 10.0
    assert numpy.array_equal(y, want)


def test_batchnorm_inference_constant_input():
    x = numpy.full((1, 2, 5), 3.0, 'f4')
    g, b = numpy.array([2., 1.], 'f4'), numpy.array([0.5, -1.], 'f4')
    m, v = numpy.array([1., 3.], 'f4'), numpy.array([4. - 2e-5, 1.], 'f4')
    y = ops.batch_norm_inference(x, g, b, m, v)
    assert numpy.allclose(y[0, 0], 2.0 * (3.0 - 1.0) / 2.0 + 0.5, atol=1e-6)
    assert numpy.allclose(y[0, 1], -1.0, atol=1e-6)


def test_activations():
    x = numpy.array([[-2., 0., 3.]], 'f4')
    assert numpy.allclose(ops.leaky_relu(x), [[-0.4, 0., 3.]])
    assert numpy.allclose(ops.relu(x), [[0., 0., 3.]])
    g = ops.glu(numpy.array([[[1.0], [0.0]]], 'f4'))                      # 1 * sigmoid(0)
    assert numpy.allclose(g, 0.5)


def test_pad_rule_is_a_full_extra_block_on_multiples():
    assert unet.pad_frames(100) == 28 and unet.pad_frames(128) == 128 and unet.pad_frames(300) == 84


@pytest.mark.parametrize('nd', [1, 2])
def test_numpy_and_torch_restatements_agree(nd):
    d = NetDesc(nd, 9 if nd == 1 else 1, 9 if nd == 1 else 1, 8, 8)
    P = synthetic_params(d, 356 + nd, bias_std=0.05)
    rng = numpy.random.default_rng(5)
    x = rng.normal(size=(1, d.in_ch, 128) if nd == 1 else (1, 1, 128, 128)).astype('f4')
    y = unet.unet_forward(x, P)
    y64 = unet.unet_forward(x.astype('f8'), {k: v.astype('f8') for k, v in P.items()})
    yt = torch_ref.TorchUNet(P).forward_np(x)
    scale = numpy.abs(y64).max()
    assert numpy.abs(y - yt).max() / scale < 2e-6
    assert numpy.abs(y - y64).max() / scale < 2e-6


C_CONV = [((2, 5, 37), (7, 5, 4), dict(stride=2, pad=1)), ((1, 6, 40), (4, 6, 3), dict(stride=1, pad=2, dilate=2)),
          ((1, 6, 41), (4, 6, 3), dict(stride=3, pad=0, dilate=2)), ((2, 3, 12, 10), (5, 3, 4, 4), dict(stride=2, pad=1)),
          ((1, 4, 9, 8), (6, 4, 3, 3), dict(stride=1, pad=1)), ((1, 4, 9, 8), (6, 4, 1, 1), dict(stride=1, pad=0)),
          ((1, 2, 7, 9), (3, 2, 3, 2), dict(stride=(2, 1), pad=(1, 0), dilate=(1, 2)))]
C_DECONV = [((2, 5, 9), (5, 7, 4), 2, 1), ((2, 3, 6, 5), (3, 4, 4, 4), 2, 1), ((1, 3, 6, 5), (3, 4, 3, 3), 1, 1), ((1, 3, 7), (3, 4, 5), 3, 2)]


@pytest.mark.parametrize('case', C_CONV, ids=lambda c: 'x'.join(map(str, c[0])))
def test_c_conv_matches_numpy_and_float64(case):
    from oracle import c_ref
    xs, ws, kw = case
    rng = numpy.random.default_rng(31)
    x, W, b = rng.normal(size=xs).astype('f4'), rng.normal(size=ws).astype('f4'), rng.normal(size=ws[0]).astype('f4')
    ref64 = ops.conv_nd(x.astype('f8'), W.astype('f8'), b.astype('f8'), **kw)
    got = c_ref.conv_nd(x, W, b, **kw)
    assert got.shape == ref64.shape and numpy.abs(got - ref64).max() < 2e-5
    assert numpy.abs(got - ops.conv_nd(x, W, b, **kw)).max() < 2e-5
    c_ref.ACC64 = True
    try:
        assert numpy.abs(c_ref.conv_nd(x, W, b, **kw) - ref64).max() < 2e-6       # double sums: only the final rounding is left
    finally:
        c_ref.ACC64 = False


@pytest.mark.parametrize('case', C_DECONV, ids=lambda c: 'x'.join(map(str, c[0])))
def test_c_deconv_matches_numpy(case):
    from oracle import c_ref
    xs, ws, st, pd = case
    rng = numpy.random.default_rng(32)
    x, W, b = rng.normal(size=xs).astype('f4'), rng.normal(size=ws).astype('f4'), rng.normal(size=ws[1]).astype('f4')
    ref64 = ops.deconv_nd(x.astype('f8'), W.astype('f8'), b.astype('f8'), stride=st, pad=pd)
    got = c_ref.deconv_nd(x, W, b, stride=st, pad=pd)
    assert got.shape == ref64.shape and numpy.abs(got - ref64).max() < 2e-5


def test_c_known_answers_and_bn():
    """The hand-written known answers of this file on the C restatement: a delta returns the taps, a one-hot through the
    k4 s2 p1 deconvolution lands the filter at 2i - 1, BatchNormalization of a constant."""
    from oracle import c_ref
    W = numpy.arange(1, 4, dtype='f4').reshape(1, 1, 3)
    x = numpy.zeros((1, 1, 7), 'f4'); x[0, 0, 3] = 1.0
    assert numpy.array_equal(c_ref.conv_nd(x, W, None, stride=1, pad=1)[0, 0], [0, 0, 3, 2, 1, 0, 0])
    Wd = numpy.arange(1, 5, dtype='f4').reshape(1, 1, 4)
    x = numpy.zeros((1, 1, 4), 'f4'); x[0, 0, 2] = 1.0
    assert numpy.array_equal(c_ref.deconv_nd(x, Wd, None, stride=2, pad=1)[0, 0], [0, 0, 0, 1, 2, 3, 4, 0])
    one = numpy.ones(2, 'f4')
    y = c_ref.batch_norm_inference(numpy.full((1, 2, 3), 5.0, 'f4'), 2 * one, 0.5 * one, 3 * one, 4 * one - 2e-5)
    assert numpy.allclose(y, 2 * (5 - 3) / 2 + 0.5)
    assert numpy.allclose(c_ref.leaky_relu(numpy.array([[[-2., 0., 3.]]], 'f4')), [[[-0.4, 0., 3.]]])
    assert numpy.allclose(c_ref.relu(numpy.array([[[-2., 0., 3.]]], 'f4')), [[[0., 0., 3.]]])


@pytest.mark.parametrize('nd', [1, 2])
def test_three_restatements_agree_on_the_predictors(nd):
    """numpy (tap-wise tensordot), torch / oneDNN and the plain-C loop nests on the same small predictor."""
    from oracle import c_ref
    d = NetDesc(nd, 9 if nd == 1 else 1, 9 if nd == 1 else 1, 8, 8)
    P = synthetic_params(d, 380 + nd, bias_std=0.05)
    x = numpy.random.default_rng(6).normal(size=(2, d.in_ch, 128) if nd == 1 else (2, 1, 128, 128)).astype('f4')
    y_np, y_c, y_t = unet.unet_forward(x, P), unet.unet_forward(x, P, ops=c_ref), torch_ref.TorchUNet(P).forward_np(x)
    scale = numpy.abs(y_np).max()
    assert numpy.abs(y_c - y_np).max() / scale < 2e-6 and numpy.abs(y_c - y_t).max() / scale < 2e-6


def test_extensive_layers_below_eight_keeps_shapes():
    d = NetDesc(1, 9, 9, 8, 3)
    P = synthetic_params(d, 1)
    x = numpy.random.default_rng(0).normal(size=(1, 9, 40)).astype('f4')
    assert unet.unet_forward(x, P, 3).shape == (1, 9, 40)


@pytest.mark.parametrize('path', sorted(glob.glob(str(GOLD / '*.npz'))), ids=lambda p: Path(p).stem)
def test_oracle_reproduces_golden_fixtures(path):
    z = numpy.load(path)
    nd, inc, outc, base, e = [int(v) for v in z['desc']]
    d = NetDesc(nd, inc, outc, base, e)
    P = synthetic_params(d, int(z['seed']), bias_std=float(z['bias_std']))
    x, y = z['x'], z['y']
    if nd == 1:
        got = unet.stage1_convert_core(x, P)
        err = numpy.abs(got - y).max() / numpy.abs(y).max()
    elif 'forward' in Path(path).stem:
        got = unet.unet_forward(x[:, None], P)[:, 0]
        err = numpy.abs(got - y).max() / numpy.abs(y).max()
    else:
        got = unet.stage2_convert(x, P)
        err = numpy.abs(got / y - 1).max()
    assert err < 1e-5, err


@pytest.mark.parametrize('path', sorted(glob.glob(str(GOLD / '*.npz'))), ids=lambda p: Path(p).stem)
def test_c_oracle_reproduces_golden_fixtures(path):
    from oracle import c_ref
    z = numpy.load(path)
    nd, inc, outc, base, e = [int(v) for v in z['desc']]
    d = NetDesc(nd, inc, outc, base, e)
    P = synthetic_params(d, int(z['seed']), bias_std=float(z['bias_std']))
    x, y = z['x'], z['y']
    if nd == 1:
        err = numpy.abs(unet.stage1_convert_core(x, P, ops=c_ref) - y).max() / numpy.abs(y).max()
    elif 'forward' in Path(path).stem:
        err = numpy.abs(unet.unet_forward(x[:, None], P, ops=c_ref)[:, 0] - y).max() / numpy.abs(y).max()
    else:
        err = numpy.abs(unet.stage2_convert(x, P, ops=c_ref) / y - 1).max()
    assert err < 1e-5, err
