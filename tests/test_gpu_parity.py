"""Parity gate: the HIP path on a real MI355X (through the C ABI) against the CPU oracle.

Tolerance: 1e-4 relative fp32 (BASELINE.json north_star), metric max|y - ref| / max|ref| for activations
and max|y / ref - 1| for the (positive) stage-2 spectrogram output.  Operator and small-predictor cases are
shared with the emulator tests (tests/cases.py); full-size cases use the canonical SYN-64 configs at
BASELINE.json's window sizes, checked against the torch/oneDNN restatement (the numpy one is too slow there)
and through size-independent properties (batch invariance, window independence, determinism)."""
import numpy
import pytest

from conftest import bn_params, rel_max
import cases
from oracle import torch_ref, unet
from realtime_yukarin_amd import engine, synth
from realtime_yukarin_amd.netspec import NetDesc
from realtime_yukarin_amd.weights import flatten_params, synthetic_params

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', cases.CONV1D_CASES, ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv1d_gpu(gpu_ctx, case):
    y, r = cases.run_conv1d(gpu_ctx, numpy.random.default_rng(11), case, bn_params)
    assert rel_max(y, r) < cases.TOL


@pytest.mark.parametrize('case', cases.CONV2D_CASES, ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv2d_gpu(gpu_ctx, case):
    y, r = cases.run_conv2d(gpu_ctx, numpy.random.default_rng(12), case, bn_params)
    assert rel_max(y, r) < cases.TOL


@pytest.mark.parametrize('case', cases.CONV2D_DILATED_CASES, ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv2d_dilated_gpu(gpu_ctx, case):
    y, r = cases.run_conv2d_dilated(gpu_ctx, numpy.random.default_rng(19), case, bn_params)
    assert y.shape == r.shape and rel_max(y, r) < cases.TOL


@pytest.mark.parametrize('case', cases.CONV2D_BF16_CASES, ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv2d_bf16_gpu(gpu_ctx, case):
    """bf16-operand MFMA (BASELINE config #5): exact up to fp32 accumulation order against the oracle run on bf16-rounded
    operands; within 2e-2 of the fp32 oracle."""
    y, r16, r32 = cases.run_conv2d_bf16(gpu_ctx, numpy.random.default_rng(16), case, bn_params)
    assert rel_max(y, r16) < 1e-4
    assert rel_max(y, r32) < 2e-2


@pytest.mark.parametrize('case', cases.CONV2D_X3_CASES, ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv2d_x3_gpu(gpu_ctx, case):
    """split-bf16 implicit GEMM (dtype 'bf16x3'): x w ~ x_hi w_hi + x_lo w_hi + x_hi w_lo on v_mfma_f32_32x32x16_bf16, fp32
    accumulate -- against the float64 model of exactly that sum and against the fp32-operand oracle (north-star bar 1e-4)."""
    y, r3, r = cases.run_conv2d_x3(gpu_ctx, numpy.random.default_rng(17), case, bn_params)
    assert rel_max(y, r3) < 1e-5
    assert rel_max(y, r) < 2e-5


@pytest.mark.parametrize('case', cases.CONV2D_OS_CASES, ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv2d_os_gpu(gpu_ctx, case):
    """ry_c2d_os (round 5): the output-stationary weight-streaming kernel on v_mfma_f32_4x4x1_16B_f32, K batched over the sixteen blocks."""
    y, r = cases.run_conv2d(gpu_ctx, numpy.random.default_rng(21), case, bn_params)
    assert rel_max(y, r) < cases.TOL


@pytest.mark.parametrize('one_round', [False, True], ids=['deconv-64-units', 'one-round'])
def test_conv2d_os_every_slice_gpu(gpu_ctx, one_round):
    """every instantiated slice of ry_c2d_os, with and without the LDS-DMA pixel path, against the implicit GEMM (round 5: the compiler's own
    timing of LDS-DMA reads let six slices read a ring slot early on the hardware only; the kernel waits explicitly since)"""
    res = cases.os_every_slice(gpu_ctx, one_round)
    assert len(res) >= (20 if one_round else 40), res
    bad = [r for r in res if not r[1] < 1e-5]
    assert not bad, bad


OS_FULL_SIZE = [          # the weight-streaming bottom of SYN-64 at the 300-frame window, planner's slice (B, H, W, Cin, Cout, k, s, p, transposed, act, path, tile, splits)
    (1, 6, 8, 512, 512, 4, 2, 1, False, 'lrelu', 'os', None, 0),          # encoder c7: 12 pixels, 16.8 MB of filters
    (1, 12, 16, 512, 512, 4, 2, 1, False, 'lrelu', 'os', None, 0),        # encoder c6: 48 pixels
    (1, 3, 4, 512, 512, 4, 2, 1, True, 'relu', 'os', None, 0),            # decoder c0
    (1, 6, 8, 1024, 512, 4, 2, 1, True, 'relu', 'os', None, 0),           # decoder c1: two sources of 512 channels, 33.5 MB
]


@pytest.mark.parametrize('case', OS_FULL_SIZE, ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv2d_os_full_size_gpu(gpu_ctx, case):
    """BASELINE layer sizes: against the implicit-GEMM path of the same operator (other summation order only: 1e-5) and, where the
    numpy oracle finishes in seconds, against the oracle; run twice: deterministic."""
    B, H, W_, Cin, Cout, k, s, p, tr, act, path, tile, splits = case
    rng = numpy.random.default_rng(31)
    x = rng.normal(size=(B, H, W_, Cin)).astype('f4')
    Wt = rng.normal(0, 0.02, size=(Cin, Cout, k, k) if tr else (Cout, Cin, k, k)).astype('f4')
    b = rng.normal(0, 0.1, Cout).astype('f4')
    bn = bn_params(rng, Cout)
    y = gpu_ctx.conv2d(x, Wt, b, bn, stride=s, pad=p, transposed=tr, act=act, path='os', tile=tile)
    y2 = gpu_ctx.conv2d(x, Wt, b, bn, stride=s, pad=p, transposed=tr, act=act, path='os', tile=tile)
    yi = gpu_ctx.conv2d(x, Wt, b, bn, stride=s, pad=p, transposed=tr, act=act, path='igemm')
    assert numpy.array_equal(y, y2)
    assert rel_max(y, yi) < 1e-5
    if H * W_ <= 12:
        xn = x.transpose(0, 3, 1, 2)
        r = cases.ops.deconv_nd(xn, Wt, b, stride=s, pad=p) if tr else cases.ops.conv_nd(xn, Wt, b, stride=s, pad=p)
        r = cases.ops.apply_act(cases.ops.batch_norm_inference(r, *bn), act).transpose(0, 2, 3, 1)
        assert rel_max(y, r) < cases.TOL


@pytest.mark.parametrize('case', cases.CONV2D_WINO_CASES, ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv2d_wino_gpu(gpu_ctx, case):
    """ry_wino_ldsdma (round 6): the k4 s2 p1 layers in Winograd F(2x2, 2x2) form on v_mfma_f32_32x32x2_f32, nine accumulator blocks per wave."""
    y, r = cases.run_conv2d(gpu_ctx, numpy.random.default_rng(23), case, bn_params)
    assert rel_max(y, r) < 1e-5


@pytest.mark.parametrize('transposed', [True, False])
def test_conv2d_wino_vs_direct_gpu(gpu_ctx, transposed):
    err, scale = cases.wino_vs_direct(gpu_ctx, transposed)
    assert err < 1e-5 and scale > 0.1, (err, scale)


@pytest.mark.parametrize('shape,transposed', [((1, 96, 128, 512, 128), True), ((1, 48, 64, 1024, 256), True), ((1, 192, 256, 128, 256), False), ((2, 96, 128, 256, 512), False)],
                         ids=['decoder_c5', 'decoder_c4', 'encoder_c2', 'encoder_c3_two_windows'])
def test_conv2d_wino_properties_full_size_gpu(gpu_ctx, shape, transposed):
    """Size-independent properties at BASELINE layer sizes (where the float64 oracle takes minutes): the Winograd operator is affine in its input, commutes
    with shifts of the image (every pixel then sits on another position of its 2 x 2 tile or on another tile) and does not depend on the external split
    beyond the summation order."""
    e_aff, e_eq, e_split = cases.wino_properties(gpu_ctx, shape, transposed)
    assert e_aff < 1e-5 and e_eq < 1e-5 and e_split < 1e-5, (e_aff, e_eq, e_split)        # (four float32 results of K = 2048 .. 4096 products each: measured 1.5e-6 .. 3.2e-6)


WINO_FULL_SIZE = [        # the eight MFMA-bound layers of SYN-64 at the 300-frame window (T = 384), the planner's plan: B, H, W, Cin, Cout, transposed
    (1, 384, 512, 64, 128, False), (1, 192, 256, 128, 256, False), (1, 96, 128, 256, 512, False), (1, 48, 64, 512, 512, False),      # encoder c1 .. c4
    (1, 24, 32, 1024, 512, True), (1, 48, 64, 1024, 256, True), (1, 96, 128, 512, 128, True), (1, 192, 256, 256, 64, True),          # decoder c3 .. c6
    (2, 48, 64, 1024, 256, True),                                                                                                     # two windows per call
]


@pytest.mark.parametrize('case', WINO_FULL_SIZE, ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv2d_wino_full_size_gpu(gpu_ctx, case):
    """BASELINE layer sizes on trained-like magnitudes (inputs behind a ReLU, filters ~ N(0, 0.02), BatchNormalization): against the direct implicit
    GEMM of the same operator (1e-5: the verdict's bar for the Winograd form), both workgroup shapes; run twice: deterministic."""
    B, H, W_, Cin, Cout, tr = case
    rng = numpy.random.default_rng(61)
    x = numpy.maximum(rng.normal(size=(B, H, W_, Cin)), 0).astype('f4')
    Wt = rng.normal(0, 0.02, size=(Cin, Cout, 4, 4) if tr else (Cout, Cin, 4, 4)).astype('f4')
    b = rng.normal(0, 0.1, Cout).astype('f4')
    bn = bn_params(rng, Cout)
    kw = dict(stride=2, pad=1, transposed=tr, act='relu' if tr else 'lrelu')
    yd = gpu_ctx.conv2d(x, Wt, b, bn, path='igemm', **kw)
    for tile in (None, (1, 0), (2, 0)):
        try:
            y = gpu_ctx.conv2d(x, Wt, b, bn, path='wino', tile=tile, **kw)
        except RuntimeError as e:                  # (no tile of that workgroup shape divides the grid: 24 x 32 has no 16-row tile of the eight-wave shape)
            assert 'no Winograd plan' in str(e) and tile == (2, 0) and (H if tr else H // 2) % 16, (case, tile, e)
            continue
        y2 = gpu_ctx.conv2d(x, Wt, b, bn, path='wino', tile=tile, **kw)
        assert numpy.array_equal(y, y2)
        assert rel_max(y, yd) < 1e-5, (case, tile, rel_max(y, yd))


def test_mfma_4x4x1_block_map_is_transpose_detecting(gpu_ctx):
    """Asymmetric 1x1 'conv' = plain GEMM with identity rows on the output-stationary path: catches a swapped row / column map of the sixteen
    4 x 4 blocks of v_mfma_f32_4x4x1_16B_f32 (registers = rows = pixels, lanes = columns = output channels) and a wrong K position of a block."""
    y, ref = cases.os_identity_rows(gpu_ctx)
    assert numpy.array_equal(y, ref)


X3_FULL_SIZE = [
    (1, 48, 64, 1024, 256, 4, 2, 1, True, 'relu', None, 0),        # decoder c4 of SYN-64 at the 300-frame window (planner's tile / splits)
    (1, 96, 128, 256, 512, 4, 2, 1, False, 'relu', None, 0),       # encoder c3 at the same window
]


@pytest.mark.parametrize('case', cases.CONV2D_X3_CASES[:2] + X3_FULL_SIZE, ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv2d_x3_power_of_two_scaling_is_exact_gpu(gpu_ctx, case):
    """Size-independent property at BASELINE layer sizes (no oracle needed): scaling the input or the filters by a power of two
    commutes with the bf16 hi / lo split and with every fp32 accumulation, so the split-bf16 result scales bit for bit."""
    y, y4, yw = cases.x3_scaling_property(gpu_ctx, numpy.random.default_rng(18), case)
    assert numpy.array_equal(y4, 4.0 * y) and numpy.array_equal(yw, y / 8.0)
    assert float(numpy.abs(y).max()) > 0.1


def test_mfma_fragment_map_is_transpose_detecting(gpu_ctx):
    """Asymmetric 1x1 'conv' = plain GEMM with A = identity rows: catches a swapped C/D row/col map."""
    Cin, Cout = 32, 128
    x = numpy.zeros((1, 4, 8, Cin), 'f4')
    for i in range(32):
        x[0, i // 8, i % 8, i] = 1.0                      # pixel i selects input channel i
    W = (numpy.arange(Cout * Cin, dtype='f4').reshape(Cout, Cin, 1, 1) % 251) / 251.0
    y = gpu_ctx.conv2d(x, W, None, None, stride=1, pad=0, path='igemm', tile='32x128', splits=1)
    ref = W[:, :, 0, 0].T                                 # y[pixel i][n] = W[n][i]
    assert numpy.array_equal(y.reshape(32, Cout), ref)


SMALL_NETS = [
    (1, 9, 9, 8, 8, 128, 1, 1), (1, 9, 9, 64, 8, 128, 1, 2), (1, 523, 9, 16, 8, 256, 1, 1), (1, 9, 9, 8, 3, 40, 1, 1),
    (2, 1, 1, 8, 8, 128, 128, 1), (2, 1, 1, 32, 8, 128, 128, 1),
    (2, 1, 1, 32, 3, 24, 40, 2), (2, 1, 1, 16, 0, 10, 12, 2), (2, 1, 1, 64, 3, 16, 16, 1),      # round 5: 2-D predictors with extensive_layers 3 / 0 ('same' 1x1 layers, k1 end layers) on the GPU
]


@pytest.mark.parametrize('cfg', SMALL_NETS, ids=lambda c: 'x'.join(str(v) for v in c))
def test_predictor_gpu(gpu_ctx, cfg):
    nd, inc, outc, base, e, T, width, B = cfg
    d = NetDesc(nd, inc, outc, base, e)
    P = synthetic_params(d, 400 + nd, bias_std=0.05)
    net = engine.Net(gpu_ctx, d, flatten_params(d, P), width=width)
    x = numpy.random.default_rng(13).normal(size=(B, T, inc if nd == 1 else width)).astype('f4')
    y = net.forward(x)
    assert rel_max(y, cases.oracle_forward(d, P, x)) < cases.TOL
    assert numpy.array_equal(y, net.forward(x)), 'graph replay must be deterministic'
    net.close()


@pytest.fixture(scope='module')
def syn64(gpu_ctx):
    (d1, P1), (d2, P2) = synth.model_params('SYN-64')
    n1 = engine.Net(gpu_ctx, d1, flatten_params(d1, P1))
    n2 = engine.Net(gpu_ctx, d2, flatten_params(d2, P2), width=synth.FFT_BINS - 1)
    yield (n1, torch_ref.TorchUNet(P1)), (n2, torch_ref.TorchUNet(P2))
    n1.close(); n2.close()


@pytest.mark.parametrize('n_frames', [100, 300, 1000, 600, 128, 1])   # BASELINE configs #3/#4, #3+extra, #2, #1, pad edge, 1 frame
def test_stage1_syn64_convert(syn64, n_frames):
    (n1, t1), _ = syn64
    x = synth.stage1_input(n_frames)[0]
    y = n1.convert(x)
    assert y.shape == (n_frames, synth.MC_DIMS)
    assert rel_max(y, torch_ref.stage1_convert_core(t1, x)) < cases.TOL


@pytest.mark.parametrize('n_frames', [100, 300, 1000])
def test_stage1_syn64_stress_input_523_channels(gpu_ctx, n_frames):
    """SURVEY.md 8(d) stress variant "mel + f0 + ap": C_in = 9 + 1 + 513 = 523 -> C_out = 9 at the full SYN-64 width (the first layer
    walks 523 input channels per lane set; the base-16 form is in SMALL_NETS)."""
    (d1, P1), _ = synth.model_params('SYN-64', stage1_in=523)
    net = engine.Net(gpu_ctx, d1, flatten_params(d1, P1))
    x = synth.stage1_input(n_frames, stress=True)[0]
    assert x.shape == (n_frames, 523)
    y = net.convert(x)
    r = torch_ref.stage1_convert_core(torch_ref.TorchUNet(P1), x)
    assert y.shape == (n_frames, synth.MC_DIMS) and rel_max(y, r) < cases.TOL
    net.close()


# BASELINE configs #3/#4 (100, 300), #5 (200, 400), and the windows whose length is already a multiple of 128: `pad = 128 - n % 128` is then
# a WHOLE extra block of 128 rows (numpy.pad mode 'minimum'), the case in which the default dead-row crop (RY_S2_CROP=2) removes the most
@pytest.mark.parametrize('n_frames', [100, 300, 128, 256, 384, 200, 400])
def test_stage2_syn64_convert(syn64, n_frames):
    _, (n2, t2) = syn64
    sp = synth.stage2_input(n_frames)[0]
    y = n2.convert(sp)
    r = torch_ref.stage2_convert(t2, sp)
    assert y.shape == r.shape == (n_frames, synth.FFT_BINS)
    assert numpy.isfinite(y).all()
    assert float(numpy.abs(y / r - 1).max()) < cases.TOL
    assert numpy.array_equal(y[:, -1], y[:, -2]), "pad(mode='edge') repeats the last predicted bin"


@pytest.mark.parametrize('wino', ['1', '0'], ids=['winograd', 'direct'])
@pytest.mark.parametrize('n_frames', [300, 100, 257, 383, 600])
def test_stage2_dead_row_crop_is_bit_identical(syn64, gpu_ctx, monkeypatch, n_frames, wino):
    """Decoder layers of a single padded window skip the rows that only feed the padding `SuperResolution.convert` crops away
    (DESIGN.md 5.1): every kept element must be bit-identical to the run that computes all padded rows, whichever layers are cropped
    (1 = only the layers it speeds up by themselves; 2 = every decoder layer the rule allows, the default)."""
    import ctypes
    _, (n2, _) = syn64
    sp = synth.stage2_input(n_frames)[0]
    reread = lambda: gpu_ctx.reload_env()
    monkeypatch.setenv('RY_WINOGRAD', wino)                            # (round 6: in Winograd form -- the default -- and with the direct kernels)
    try:
        out, out3 = {}, []
        for mode in ('0', '1', '2'):
            monkeypatch.setenv('RY_S2_CROP', mode); reread()
            n2.set_dtype('f32')                                         # drops the launch plans and the graphs captured under the previous setting
            out[mode] = n2.convert(sp)                                  # launch by launch
            out[mode + 'g'] = [n2.convert(sp) for _ in range(2)][-1]    # graph replay
            if n_frames in (100, 300):                                  # three windows in one call: a row prefix of every image
                out3.append(n2.convert(numpy.stack([sp, sp[::-1], sp])))
        for k in out:
            assert numpy.array_equal(out[k], out['0']), (n_frames, k)
        for o in out3:
            assert numpy.array_equal(o, out3[0]) and numpy.array_equal(o[2], o[0])
            assert float(numpy.abs(o[0] / out['0'] - 1).max()) < 1e-5          # a batch may run under another plan: summation order only
    finally:
        monkeypatch.delenv('RY_S2_CROP', raising=False); monkeypatch.delenv('RY_WINOGRAD', raising=False)
        reread(); n2.set_dtype('f32')


@pytest.mark.parametrize('wino', ['1', '0'], ids=['winograd', 'direct'])
@pytest.mark.parametrize('n_frames', [300, 100, 257, 130, 600])
def test_stage2_identical_padding_rows_are_copied_bit_identical(syn64, gpu_ctx, monkeypatch, n_frames, wino):
    """Behind the real frames the padded window is copies of one row, so every encoder layer has output rows that are equal bit for bit; the implicit
    GEMM leaves whole tile rows of that stretch out of its grid and ry_rep_rows copies the row above them (RY_S2_HOLE, default on): bit-identical to
    the run that computes them -- launch by launch, under graph replay, three windows per call -- and the grids of encoder c1 / c2 shrink."""
    import ctypes
    _, (n2, _) = syn64
    sp = synth.stage2_input(n_frames)[0]
    reread = lambda: gpu_ctx.reload_env()
    monkeypatch.setenv('RY_WINOGRAD', wino)                            # (round 6: in Winograd form -- the default -- and with the direct kernels)
    try:
        out, out3, grids = {}, {}, {}
        for mode in ('0', '1'):
            monkeypatch.setenv('RY_S2_HOLE', mode); reread()
            n2.set_dtype('f32')
            out[mode] = n2.convert(sp)
            out[mode + 'g'] = [n2.convert(sp) for _ in range(2)][-1]
            out3[mode] = n2.convert(numpy.stack([sp, sp[::-1], sp]))
            grids[mode] = {(q['layer'], q['name'].split('<')[0]): q['grid'][0] for q in n2.profile(1, n_frames, 1, window=True)}
        for k in out:
            assert numpy.array_equal(out[k], out['0']), (n_frames, k)
        assert numpy.array_equal(out3['0'], out3['1'])
        assert not [k for k in grids['0'] if k[1] == 'ry_rep_rows']
        for dtype in ('bf16x3', 'bf16'):          # the bf16 pipes: the copies the consumers read ([pixel][N] bf16, split [pixel][hi | lo]) are filled in too
            o16 = {}
            for mode in ('0', '1'):
                monkeypatch.setenv('RY_S2_HOLE', mode); reread()
                n2.set_dtype(dtype)
                o16[mode] = n2.convert(sp)
                if n_frames == 300 and mode == '1':
                    assert [q for q in n2.profile(1, n_frames, 1, window=True) if q['name'] == 'ry_rep_rows'], dtype
            assert numpy.array_equal(o16['0'], o16['1']), (n_frames, dtype)
        if n_frames == 300:             # 40 of 192 rows of encoder c1 are identical: five tile rows of six rows (30 rows) go; 19 of 96 of c2: two tile rows
            assert grids['1'][('encoder/c1', 'ry_rep_rows')] > 0 and grids['1'][('encoder/c2', 'ry_rep_rows')] > 0
            if wino == '0':
                assert grids['1'][('encoder/c1', 'ry_igemm_ldsdma')] == 432 and grids['0'][('encoder/c1', 'ry_igemm_ldsdma')] == 512
                assert grids['1'][('encoder/c2', 'ry_igemm_ldsdma')] == 224 and grids['0'][('encoder/c2', 'ry_igemm_ldsdma')] == 256
            else:                       # 8-row Winograd tiles: 32 of the 40 identical rows of encoder c1, 16 of the 19 of c2 are left out of the grids
                for layer in ('encoder/c1', 'encoder/c2'):
                    assert grids['1'][(layer, 'ry_wino_ldsdma')] < grids['0'][(layer, 'ry_wino_ldsdma')], (layer, grids)
    finally:
        monkeypatch.delenv('RY_S2_HOLE', raising=False); monkeypatch.delenv('RY_WINOGRAD', raising=False)
        reread(); n2.set_dtype('f32')


@pytest.mark.parametrize('n_frames,discard', [(300, (100, 100)), (300, (0, 37)), (300, (150, 0)), (100, (20, 20)), (600, (200, 200)), (257, (1, 1))])
def test_stage2_discarded_frames_are_not_computed(syn64, n_frames, discard):
    """`ry_sr_convert_rows` (the frames a caller like ConvertStream.process throws away are announced): kept rows bit-identical to the full
    call -- launch by launch, under graph replay, and for three windows per call --, discarded rows zero, and the decoder grids shrink."""
    _, (n2, _) = syn64
    sp = synth.stage2_input(n_frames)[0]
    full = n2.convert(sp)
    k0, k1 = discard[0], n_frames - discard[1]
    for _ in range(3):                                                    # eager, then the captured graph
        part = n2.convert(sp, discard=discard)
        assert numpy.array_equal(part[k0:k1], full[k0:k1])
        assert not part[:k0].any() and not part[k1:].any()
    three = n2.convert(numpy.stack([sp, sp[::-1], sp]), discard=discard)
    assert numpy.array_equal(three[0], three[2]) and float(numpy.abs(three[0][k0:k1] / full[k0:k1] - 1).max()) < 1e-5
    assert not three[1][:k0].any() and not three[1][k1:].any()
    assert numpy.array_equal(n2.convert(sp), full)                        # the full call afterwards is the full call


def test_stage2_syn64_against_the_c_restatement(syn64):
    """Full-size stage 2 against the plain-C loop nests (oracle/ops_ref.c), float and double sums: the HIP path sits as close
    to the double-sum result as the fp32 CPU restatement does."""
    from oracle import c_ref
    _, (n2, _) = syn64
    P2 = synth.model_params('SYN-64')[1][1]
    sp = synth.stage2_input(100)[0]
    y = n2.convert(sp)
    r32 = unet.stage2_convert(sp, P2, ops=c_ref)
    c_ref.ACC64 = True
    try:
        r64 = unet.stage2_convert(sp, P2, ops=c_ref)
    finally:
        c_ref.ACC64 = False
    e_gpu, e_cpu = float(numpy.abs(y / r64 - 1).max()), float(numpy.abs(r32 / r64 - 1).max())
    print('stage-2 vs double-sum C oracle: HIP %.2e, fp32 C %.2e' % (e_gpu, e_cpu))
    assert float(numpy.abs(y / r32 - 1).max()) < cases.TOL and e_gpu < cases.TOL
    assert e_gpu < 10 * e_cpu + 1e-6


def test_windows_are_independent_and_batch_invariant(syn64):
    """Chunk-parallel property: a window's result does not depend on its batch neighbours."""
    (n1, _), (n2, _) = syn64
    x = synth.stage1_input(100, windows=3)
    sp = synth.stage2_input(100, windows=2)
    yb = n1.convert(x)
    for w in range(3):
        assert rel_max(yb[w], n1.convert(x[w])) < 1e-6
    sb = n2.convert(sp)
    for w in range(2):
        assert float(numpy.abs(sb[w] / n2.convert(sp[w]) - 1).max()) < 1e-5


def test_stage2_syn64_forward_400_frames(syn64):
    """BASELINE config #5 window (N=400 -> 512 padded frames), raw predictor."""
    _, (n2, t2) = syn64
    x = numpy.log(synth.stage2_input(512)[:, :, :-1])
    y = n2.forward(x)
    r = t2.forward_np(x[:, numpy.newaxis])[:, 0]
    assert rel_max(y, r) < cases.TOL


BF16_TOL = 3e-2     # stated tolerance of the bf16 stage-2 variant against the fp32 oracle, on the log-spectrum (max / max)


def test_stage2_syn64_bf16_variant(gpu_ctx):
    """BASELINE config #5 (buffer_time 1.0 s -> N = 200 core / 400 with extra): bf16 operands, fp32 accumulate."""
    (_, _), (d2, P2) = synth.model_params('SYN-64')
    net = engine.Net(gpu_ctx, d2, flatten_params(d2, P2), width=synth.FFT_BINS - 1)
    t2 = torch_ref.TorchUNet(P2)
    sp = synth.stage2_input(200)[0]
    y32 = net.convert(sp)
    net.set_dtype('bf16')
    y16 = net.convert(sp)
    net.set_dtype('f32')
    assert numpy.array_equal(net.convert(sp), y32), 'switching back restores the exact fp32 path'
    r = torch_ref.stage2_convert(t2, sp)
    e32 = rel_max(numpy.log(y32), numpy.log(r))
    e16 = rel_max(numpy.log(y16), numpy.log(r))
    print('stage-2 log-spectrum error vs fp32 oracle: fp32 path %.2e, bf16 path %.2e' % (e32, e16))
    assert e32 < cases.TOL
    assert 1e-5 < e16 < BF16_TOL, e16
    net.close()


def test_stage2_syn64_x3_variant(gpu_ctx):
    """split-bf16 ('bf16x3') stage-2 at the BASELINE config #3 window (300 frames): three bf16 products per fp32 product on the
    bf16 matrix pipe.  The mode has to stay inside the SAME parity bar as the fp32 path (1e-4, cases.TOL); measured ~2e-6."""
    (_, _), (d2, P2) = synth.model_params('SYN-64')
    net = engine.Net(gpu_ctx, d2, flatten_params(d2, P2), width=synth.FFT_BINS - 1)
    t2 = torch_ref.TorchUNet(P2)
    sp = synth.stage2_input(300)[0]
    y32 = net.convert(sp)
    net.set_dtype('bf16x3')
    y3 = net.convert(sp)
    names = [q['name'] for q in net.profile(1, 384, 1)]
    if True:
        assert sum(n.startswith('ry_igemm_ldsdma<') and n[:-1].split(',')[5] == 'true' for n in names) >= 8, names   # the MFMA-bound layers did take the bf16 pipe
    net.set_dtype('f32')
    assert numpy.array_equal(net.convert(sp), y32), 'switching back restores the exact fp32 path'
    r = torch_ref.stage2_convert(t2, sp)
    e32 = rel_max(numpy.log(y32), numpy.log(r))
    e3 = rel_max(numpy.log(y3), numpy.log(r))
    print('stage-2 log-spectrum error vs fp32 oracle: fp32 path %.2e, split-bf16 path %.2e; sp max rel %.2e' % (e32, e3, float(numpy.abs(y3 / r - 1).max())))
    assert e32 < cases.TOL
    assert e3 < 2e-5, e3                   # 5 x under the 1e-4 bar
    assert float(numpy.abs(y3 / r - 1).max()) < cases.TOL
    net.close()


def test_device_resident_voice_changer_core(syn64):
    """stage-1 -> combine_silent -> mc2sp -> +1e-16 -> stage-2 in one `ry_vc_convert` against the
    step-by-step composition: torch oracle CNNs + the independent restatement of pysptk.mc2sp (oracle/mc2sp.py, float64)."""
    from oracle import mc2sp as omc
    from realtime_yukarin_amd import sptk
    (n1, t1), (n2, t2) = syn64
    n = 300
    rng = numpy.random.default_rng(99)
    effective = rng.random(n) > 0.25
    x = synth.stage1_input(n)[0]
    alpha = sptk.mcepalpha(16000)
    core = engine.VcCore(n1, n2, sptk.mc2sp_matrix(8, alpha, 1024))
    mc, sp = core.convert(x[effective], effective)
    mc_ref = numpy.zeros((n, synth.MC_DIMS), numpy.float32)
    mc_ref[effective] = torch_ref.stage1_convert_core(t1, x[effective])
    assert rel_max(mc, mc_ref) < cases.TOL and not mc[~effective].any()
    sp_mid = (omc.mc2sp(mc_ref, omc.mcepalpha(16000), 1024) + 1e-16).astype(numpy.float32)
    sp_ref = torch_ref.stage2_convert(t2, sp_mid)
    assert float(numpy.abs(sp / sp_ref - 1).max()) < cases.TOL
    # repeated calls with a varying number of effective frames (graph re-use / eager switching) stay consistent
    for keep in (0.9, 0.5, 0.9, 0.9):
        eff = rng.random(n) < keep
        mc2, _ = core.convert(x[eff], eff)
        ref2 = numpy.zeros_like(mc_ref); ref2[eff] = torch_ref.stage1_convert_core(t1, x[eff])
        assert rel_max(mc2, ref2) < cases.TOL
    # the ragged ends: no effective frame at all (voice_changer.py:32-35: the stage-1 CNN is not called, mc stays the all-silent zeros and
    # stage 2 converts mc2sp(0) + 1e-16), and a single effective frame (stage 1 pads 1 -> 128 with that frame)
    none = numpy.zeros(n, bool)
    mc0, sp0 = core.convert(x[none], none)
    flat = (omc.mc2sp(numpy.zeros((n, synth.MC_DIMS)), omc.mcepalpha(16000), 1024) + 1e-16).astype(numpy.float32)
    assert not mc0.any() and float(numpy.abs(sp0 / torch_ref.stage2_convert(t2, flat) - 1).max()) < cases.TOL
    lone = none.copy(); lone[137] = True
    mc1, sp1 = core.convert(x[lone], lone)
    ref1 = numpy.zeros_like(mc_ref); ref1[lone] = torch_ref.stage1_convert_core(t1, x[lone])
    assert rel_max(mc1, ref1) < cases.TOL and not mc1[~lone].any()
    sp1_ref = torch_ref.stage2_convert(t2, (omc.mc2sp(ref1, omc.mcepalpha(16000), 1024) + 1e-16).astype(numpy.float32))
    assert float(numpy.abs(sp1 / sp1_ref - 1).max()) < cases.TOL
    core.close()


def test_window_call_lanes_discard_and_batch(syn64):
    """The window call on the real GPU: windows in flight over the two lanes (clones of the predictor pair) return the same bits as one
    lane; `ry_vc_set_discard` leaves every kept row of the spectrogram bit-identical (host ring and device-pointer call alike), zeros in
    the discarded rows, mc complete; the batch call agrees with the windows one by one."""
    from realtime_yukarin_amd import sptk
    (n1, _), (n2, _) = syn64
    n = 300
    rng = numpy.random.default_rng(77)
    wins = []
    for i in range(5):
        e = rng.random(n) > (0.0 if i % 2 else 0.3)
        wins.append((synth.stage1_input(n, seed=500 + i)[0][e], e))
    mtx = sptk.mc2sp_matrix(8, sptk.mcepalpha(16000), 1024)
    one = engine.VcCore(n1, n2, mtx, lanes=1)
    ref = [one.convert(x, e) for x, e in wins]
    one.close()
    core = engine.VcCore(n1, n2, mtx)                                           # default: two lanes
    assert core.lanes == 2
    for rep in range(2):                                                        # eager, then graph replay, six in flight
        got = list(core.convert_stream(wins + wins[:1], depth=6))
        for (mc, sp), (rmc, rsp) in zip(got, ref + ref[:1]):
            assert numpy.array_equal(mc, rmc) and numpy.array_equal(sp, rsp)
    core.set_discard(100, 100)
    for (x, e), (rmc, rsp) in zip(wins[:3], ref):
        mc, sp = core.convert(x, e)
        assert numpy.array_equal(sp[100:200], rsp[100:200]) and not sp[:100].any() and not sp[200:].any() and numpy.array_equal(mc, rmc)
    batch = core.convert_batch(wins[:3])
    for (mc, sp), (rmc, rsp) in zip(batch, ref):
        assert float(numpy.abs(sp[100:200] / rsp[100:200] - 1).max()) < 1e-5 and not sp[:100].any() and not sp[200:].any()
    core.set_discard(0, 0)
    batch = core.convert_batch(wins[:3])
    for (mc, sp), (rmc, rsp) in zip(batch, ref):
        assert float(numpy.abs(sp / rsp - 1).max()) < 1e-5 and float(numpy.abs(mc - rmc).max()) <= 1e-5 * float(numpy.abs(rmc).max())
    mc, sp = core.convert(*wins[0])
    assert numpy.array_equal(sp, ref[0][1])
    core.set_lanes(4)                                                           # four lanes, eight ring slots: still the same bits
    assert core.ring == 8
    got = list(core.convert_stream(wins + wins, depth=8))
    for (mc, sp), (rmc, rsp) in zip(got, ref + ref):
        assert numpy.array_equal(mc, rmc) and numpy.array_equal(sp, rsp)
    core.close()


def test_errors_are_reported_not_fatal(gpu_ctx):
    d = NetDesc(1, 9, 9, 8, 8)
    P = synthetic_params(d, 1)
    net = engine.Net(gpu_ctx, d, flatten_params(d, P))
    with pytest.raises(Exception) as ei:
        net.forward(numpy.zeros((1, 100, 9), 'f4'))       # 100 is not a multiple of 128
    assert 'multiple' in str(ei.value)
    assert net.forward(numpy.zeros((1, 128, 9), 'f4')).shape == (1, 128, 9)   # still usable afterwards
    net.close()


def _chained_oracle(t1, t2, x, effective, n):
    """voice_changer.py:33-41 on the oracle for one window: stage 1 on the effective rows, zeros elsewhere, mc2sp + 1e-16, stage 2."""
    from oracle import mc2sp as omc
    mc = numpy.zeros((n, synth.MC_DIMS), numpy.float32)
    if effective.any():
        mc[effective] = torch_ref.stage1_convert_core(t1, x)
    sp_mid = (omc.mc2sp(mc, omc.mcepalpha(16000), 1024) + 1e-16).astype(numpy.float32)
    return mc, torch_ref.stage2_convert(t2, sp_mid)


def test_no_kernel_reads_what_its_producer_did_not_write_gpu(syn64, gpu_ctx, monkeypatch):
    """BASELINE window sizes, all three arithmetic modes, with and without a discard: poisoned activation buffers (RY_POISON) leave no NaN."""
    (_, _), (n2, _) = syn64
    sizes = [(n, synth.stage2_input(n, seed=950 + n)[0]) for n in (300, 100, 400, 128)]
    res = cases.poisoned_converts(gpu_ctx, n2, sizes, monkeypatch)
    assert all(r[2] == 0 and r[3] == 0 for r in res), res


def test_config5_bf16_through_the_chained_core_at_400_frames(syn64):
    """BASELINE config #5 as BASELINE.json words it -- bf16 stage-2 on the matrix pipe, buffer_time 1.0 s (+ 2 x 0.5 s extra: N = 400 -> 512
    padded frames) -- through the CHAINED window call (stage-1 fp32 -> combine_silent -> mc2sp -> stage-2 bf16), not the raw stage-2
    convert: mc stays inside the fp32 bar, the spectrogram inside the stated 3e-2 on the log-spectrum, against the fp32 oracle."""
    from realtime_yukarin_amd import sptk
    (n1, t1), (n2, t2) = syn64
    n = 400
    rng = numpy.random.default_rng(405)
    effective = rng.random(n) > 0.2
    x = synth.stage1_input(n, seed=406)[0][effective]
    mc_ref, sp_ref = _chained_oracle(t1, t2, x, effective, n)
    core = engine.VcCore(n1, n2, sptk.mc2sp_matrix(8, sptk.mcepalpha(16000), 1024))
    try:
        n2.set_dtype('bf16')
        got = list(core.convert_stream([(x, effective)] * 3, depth=3))            # both lanes (the clones follow the handle's mode), eager then graph replay
    finally:
        n2.set_dtype('f32')
    mc32, sp32 = core.convert(x, effective)
    core.close()
    for mc, sp in got:
        e16 = rel_max(numpy.log(sp), numpy.log(sp_ref))
        assert rel_max(mc, mc_ref) < cases.TOL and not mc[~effective].any()
        assert 1e-5 < e16 < BF16_TOL, e16
        if not numpy.array_equal(sp, got[0][1]):                                  # every lane, eager or replayed: the same bits
            dd = numpy.argwhere(sp != got[0][1])
            raise AssertionError('windows differ: %d elements, rows %s .. %s, cols %s, max rel %.3g; windows equal to the first: %s' % (
                len(dd), sorted(set(dd[:, 0].tolist()))[:8], sorted(set(dd[:, 0].tolist()))[-4:], sorted(set(dd[:, 1].tolist()))[:8],
                float(numpy.abs(sp / got[0][1] - 1).max()), [bool(numpy.array_equal(g[1], got[0][1])) for g in got]))
    print('config #5 chained, 400 frames: bf16 log-spectrum error %.2e (stated %.0e); fp32 path element-wise %.2e'
          % (rel_max(numpy.log(got[0][1]), numpy.log(sp_ref)), BF16_TOL, float(numpy.abs(sp32 / sp_ref - 1).max())))
    assert float(numpy.abs(sp32 / sp_ref - 1).max()) < cases.TOL                # back in f32 mode: the exact path again


def test_convert_batch_of_8_windows_against_the_oracle(syn64):
    """`VcCore.convert_batch` / `ry_vc_enqueue_device_batch` -- eight different 300-frame windows in one call, two of them cut by the
    silence gate (so stage 1 runs window by window) -- each window against the ORACLE directly (round 3 compared the batch call with
    the single-window call only); then eight all-effective windows (stage 1 as one batch)."""
    from realtime_yukarin_amd import sptk
    (n1, t1), (n2, t2) = syn64
    n = 300
    rng = numpy.random.default_rng(808)
    core = engine.VcCore(n1, n2, sptk.mc2sp_matrix(8, sptk.mcepalpha(16000), 1024))
    for gated in ((2, 5), ()):
        wins = []
        for w in range(8):
            e = numpy.ones(n, bool)
            if w in gated:
                e[40 * w:40 * w + 60] = False
            wins.append((synth.stage1_input(n, seed=810 + w)[0][e], e))
        res = core.convert_batch(wins)
        worst_sp = worst_mc = 0.0
        for (x, e), (mc, sp) in zip(wins, res):
            mc_ref, sp_ref = _chained_oracle(t1, t2, x, e, n)
            worst_mc = max(worst_mc, rel_max(mc, mc_ref)); worst_sp = max(worst_sp, float(numpy.abs(sp.astype(numpy.float64) / sp_ref - 1).max()))
            assert not mc[~e].any()
        print('convert_batch of 8 (gated windows: %s): worst mc %.2e, worst sp element-wise %.2e' % (list(gated), worst_mc, worst_sp))
        assert worst_mc < cases.TOL and worst_sp < cases.TOL
    core.close()


def test_long_windows_2000_frames(syn64):
    """The far end of the window sizes: 2000 real frames (2048 padded; 10 s of audio in one window -- an offline caller's choice, five
    times BASELINE config #2) through the chained window call against the oracle, with a silent stretch; then the same window cut to
    its last frame before a pad boundary (1919 -> 1920 padded) so that both 128-multiples around it are exercised."""
    from realtime_yukarin_amd import sptk
    (n1, t1), (n2, t2) = syn64
    core = engine.VcCore(n1, n2, sptk.mc2sp_matrix(8, sptk.mcepalpha(16000), 1024))
    for n in (2000, 1919):
        effective = numpy.ones(n, bool); effective[700:1100] = False
        x = synth.stage1_input(n, seed=2000 + n)[0][effective]
        mc, sp = core.convert(x, effective)
        mc_ref, sp_ref = _chained_oracle(t1, t2, x, effective, n)
        e_mc, e_sp = rel_max(mc, mc_ref), float(numpy.abs(sp.astype(numpy.float64) / sp_ref - 1).max())
        print('chained window of %d frames: mc %.2e, sp element-wise %.2e' % (n, e_mc, e_sp))
        assert mc.shape == (n, synth.MC_DIMS) and sp.shape == (n, synth.FFT_BINS)
        assert e_mc < cases.TOL and e_sp < cases.TOL and not mc[~effective].any()
    core.close()
