"""The gfx950 kernel sources, compiled for the host-side SIMT emulator, against the oracle (CPU, no GPU).

This checks index arithmetic, MFMA fragment maps (as documented), split-K, deferred epilogues, the
wrapper kernels and the executor.  It is NOT the parity gate (tests/test_gpu_parity.py is); it exists so
kernel logic can be iterated where no MI355X is attached."""
import numpy
import pytest

from conftest import bn_params, rel_max
import cases
from oracle import unet
from realtime_yukarin_amd import engine
from realtime_yukarin_amd.netspec import NetDesc
from realtime_yukarin_amd.weights import flatten_params, synthetic_params


@pytest.mark.parametrize('case', cases.CONV1D_CASES, ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv1d_emu(emu_ctx, case):
    y, r = cases.run_conv1d(emu_ctx, numpy.random.default_rng(11), case, bn_params)
    assert rel_max(y, r) < cases.TOL


@pytest.mark.parametrize('case', cases.CONV2D_CASES, ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv2d_emu(emu_ctx, case):
    y, r = cases.run_conv2d(emu_ctx, numpy.random.default_rng(12), case, bn_params)
    assert rel_max(y, r) < cases.TOL


@pytest.mark.parametrize('case', cases.CONV2D_OS_CASES, ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv2d_os_emu(emu_ctx, case):
    """the output-stationary weight-streaming kernel (ry_c2d_os): emulated 4x4x1 block map, K runs per wave, reduce-scatter over the blocks"""
    y, r = cases.run_conv2d(emu_ctx, numpy.random.default_rng(21), case, bn_params)
    assert rel_max(y, r) < cases.TOL


def test_conv2d_os_every_slice_emu(emu_ctx):
    res = cases.os_every_slice(emu_ctx, one_round=True)
    assert len(res) >= 20 and all(e < 1e-5 for _, e in res), res


def test_conv2d_os_identity_rows_emu(emu_ctx):
    y, ref = cases.os_identity_rows(emu_ctx)
    assert numpy.array_equal(y, ref)


@pytest.mark.parametrize('case', cases.CONV2D_WINO_CASES, ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv2d_wino_emu(emu_ctx, case):
    """the Winograd F(2x2, 2x2) kernel (ry_wino_ldsdma): patch layout + swizzle, in-register input transform, nine accumulator blocks, output transform"""
    y, r = cases.run_conv2d(emu_ctx, numpy.random.default_rng(23), case, bn_params)
    assert rel_max(y, r) < 1e-5


@pytest.mark.parametrize('transposed', [True, False])
def test_conv2d_wino_vs_direct_emu(emu_ctx, transposed):
    err, scale = cases.wino_vs_direct(emu_ctx, transposed)
    assert err < 1e-5 and scale > 0.1, (err, scale)


@pytest.mark.parametrize('transposed', [True, False])
def test_conv2d_wino_properties_emu(emu_ctx, transposed):
    """affinity, shift equivariance and split invariance of the Winograd operator (small size; `-m gpu` runs the BASELINE layer sizes)"""
    shape = (1, 16, 32, 48, 64) if transposed else (1, 32, 64, 32, 64)
    e = cases.wino_properties(emu_ctx, shape, transposed)
    assert max(e) < 2e-6, e


@pytest.mark.parametrize('case', cases.CONV2D_DILATED_CASES, ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv2d_dilated_emu(emu_ctx, case):
    y, r = cases.run_conv2d_dilated(emu_ctx, numpy.random.default_rng(19), case, bn_params)
    assert y.shape == r.shape and rel_max(y, r) < cases.TOL


@pytest.mark.parametrize('case', cases.CONV2D_BF16_CASES, ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv2d_bf16_emu(emu_ctx, case):
    y, r16, r32 = cases.run_conv2d_bf16(emu_ctx, numpy.random.default_rng(16), case, bn_params)
    assert rel_max(y, r16) < 1e-5          # exact model of the kernel: bf16-rounded operands, fp32 accumulation
    assert rel_max(y, r32) < 2e-2          # and the price of bf16 operands against the fp32 oracle


@pytest.mark.parametrize('case', cases.CONV2D_X3_CASES, ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv2d_x3_emu(emu_ctx, case):
    """split-bf16 implicit GEMM: [hi | lo] sources, K axis [hi | lo | hi] against filters [hi | hi | lo]."""
    y, r3, r = cases.run_conv2d_x3(emu_ctx, numpy.random.default_rng(17), case, bn_params)
    assert rel_max(y, r3) < 1e-5           # exact model of the kernel up to fp32 accumulation order (K is 3 x 16 x Cin terms)
    assert rel_max(y, r) < 1e-5            # and against the fp32-operand oracle: the dropped lo*lo term and the 16-bit split


@pytest.mark.parametrize('case', cases.CONV2D_X3_CASES[:3], ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv2d_x3_power_of_two_scaling_is_exact_emu(emu_ctx, case):
    y, y4, yw = cases.x3_scaling_property(emu_ctx, numpy.random.default_rng(18), case)
    assert numpy.array_equal(y4, 4.0 * y) and numpy.array_equal(yw, y / 8.0)
    assert float(numpy.abs(y).max()) > 0.1


NETS = [
    # ndim, in, out, base, e, T, width, batch
    (1, 9, 9, 8, 8, 128, 1, 1),
    (1, 9, 9, 64, 8, 128, 1, 2),
    (1, 523, 9, 16, 8, 256, 1, 1),      # "mel+f0+ap" stress variant
    (1, 9, 9, 8, 3, 40, 1, 1),          # extensive_layers 3: 'same' layers
    (2, 1, 1, 8, 8, 128, 128, 1),       # all-direct path
    (2, 1, 1, 32, 8, 128, 128, 1),      # implicit-GEMM middle layers
    (2, 1, 1, 32, 3, 24, 40, 2),        # 2-D, extensive_layers 3: two down / up layers, the rest 'same' 1x1 layers; batch 2, ragged sizes
    (2, 1, 1, 16, 0, 10, 12, 2),        # 2-D, extensive_layers 0: every layer 1x1 (k1 end layers)
]


@pytest.mark.parametrize('cfg', NETS, ids=lambda c: 'x'.join(str(v) for v in c))
def test_predictor_emu(emu_ctx, cfg):
    nd, inc, outc, base, e, T, width, B = cfg
    d = NetDesc(nd, inc, outc, base, e)
    P = synthetic_params(d, 400 + nd, bias_std=0.05)
    net = engine.Net(emu_ctx, d, flatten_params(d, P), width=width)
    rng = numpy.random.default_rng(13)
    x = rng.normal(size=(B, T, inc if nd == 1 else width)).astype('f4')
    assert rel_max(net.forward(x), cases.oracle_forward(d, P, x)) < cases.TOL
    net.close()


def test_stage2_bf16_pipeline_emu(emu_ctx):
    """bf16 mode of the stage-2 predictor: implicit-GEMM layers read / write bf16 activations (the first layer and the
    split-K reduce kernels write bf16 copies), the end layers stay fp32.  A base-64 net with two stride-2 levels keeps the
    emulator run short.  Stated tolerance of the mode against the fp32 oracle: 3e-2 (tests/test_gpu_parity.py)."""
    d = NetDesc(2, 1, 1, 64, 2)
    P = synthetic_params(d, 420, bias_std=0.05)
    net = engine.Net(emu_ctx, d, flatten_params(d, P), width=16)
    x = numpy.random.default_rng(21).normal(size=(1, 8, 16)).astype('f4')
    ref = cases.oracle_forward(d, P, x)
    y32 = net.forward(x)
    assert rel_max(y32, ref) < cases.TOL
    net.set_dtype('bf16')
    y16 = net.forward(x)
    names = [q['name'] for q in net.profile(1, 8, 1)]
    assert any(n.startswith('ry_igemm_ldsdma<') and n[:-1].split(',')[5] == 'true' for n in names), names   # the bf16 kernels did run
    assert not numpy.array_equal(y16, y32)
    assert rel_max(y16, ref) < 3e-2
    net.set_dtype('f32')
    assert numpy.array_equal(net.forward(x), y32)                      # and the exact path comes back bit for bit
    net.close()


@pytest.mark.parametrize('n_frames', [1, 37, 100, 128])
def test_stage1_convert_emu(emu_ctx, n_frames):
    d = NetDesc(1, 9, 9, 8, 8)
    P = synthetic_params(d, 410, bias_std=0.05)
    net = engine.Net(emu_ctx, d, flatten_params(d, P))
    x = numpy.random.default_rng(14).normal(size=(n_frames, 9)).astype('f4')
    y = net.convert(x)
    assert y.shape == (n_frames, 9)
    assert rel_max(y, unet.stage1_convert_core(x, P)) < cases.TOL
    net.close()


@pytest.mark.parametrize('n_frames', [5, 100])
def test_stage2_convert_emu(emu_ctx, n_frames):
    d = NetDesc(2, 1, 1, 8, 8)
    P = synthetic_params(d, 411, bias_std=0.05)
    net = engine.Net(emu_ctx, d, flatten_params(d, P), width=128)
    sp = (numpy.exp(numpy.random.default_rng(15).normal(-6, 1.5, size=(2, n_frames, 129))) + 1e-16).astype('f4')
    y = net.convert(sp)                                   # two windows in one call
    for w in range(2):
        r = unet.stage2_convert(sp[w], P)
        assert float(numpy.abs(y[w] / r - 1).max()) < cases.TOL
    net.close()


def test_convert_edge_inputs_emu(emu_ctx):
    """Edges of the two wrappers: an empty window is an error (numpy.pad mode='minimum' refuses an empty axis in the reference's
    wrappers too), a wrong bin count is refused before any launch, non-finite input propagates instead of being masked, and
    other float dtypes are cast like `.astype(numpy.float32)` at voice_changer.py:41."""
    d1, d2 = NetDesc(1, 9, 9, 8, 8), NetDesc(2, 1, 1, 8, 8)
    n1 = engine.Net(emu_ctx, d1, flatten_params(d1, synthetic_params(d1, 5)))
    n2 = engine.Net(emu_ctx, d2, flatten_params(d2, synthetic_params(d2, 6)), width=128)
    with pytest.raises(Exception, match='positive'):
        n1.convert(numpy.zeros((0, 9), 'f4'))
    with pytest.raises(Exception, match='positive'):
        n2.convert(numpy.zeros((0, 129), 'f4'))
    with pytest.raises(ValueError, match='bins'):
        n2.convert(numpy.ones((5, 128), 'f4'))
    x64 = numpy.random.default_rng(1).normal(size=(7, 9))
    assert numpy.array_equal(n1.convert(x64), n1.convert(x64.astype('f4')))
    sp = numpy.full((5, 129), 1e-3, 'f4')
    sp[2, 40] = numpy.nan
    assert numpy.isnan(n2.convert(sp)).any()
    assert numpy.isfinite(n2.convert(numpy.full((5, 129), 1e-3, 'f4'))).all()      # ... and the net is still usable afterwards
    n1.close(); n2.close()


def test_stage2_x3_pipeline_emu(emu_ctx, monkeypatch):
    """split-bf16 ('bf16x3') mode of the stage-2 predictor: implicit-GEMM layers read [hi | lo] bf16 activations written by
    their producers (first layer, implicit-GEMM epilogue, split-K reduce) and run three bf16 products per fp32 product.
    Results must stay inside the fp32 parity bar (1e-4); layers under RY_X3_MINM rows stay on the exact fp32 kernel."""
    d = NetDesc(2, 1, 1, 64, 3)
    P = synthetic_params(d, 421, bias_std=0.05)
    net = engine.Net(emu_ctx, d, flatten_params(d, P), width=16)
    x = numpy.random.default_rng(22).normal(size=(1, 16, 16)).astype('f4')
    ref = cases.oracle_forward(d, P, x)
    y32 = net.forward(x)
    assert rel_max(y32, ref) < cases.TOL
    monkeypatch.setenv('RY_X3_MINM', '32')            # encoder c1 (8 x 8 rows) and decoder c5 / c6 split, encoder c2 (4 x 4) stays fp32
    net.set_dtype('bf16x3')
    y3 = net.forward(x)
    st = net.profile(1, 16, 1)
    split = [q['layer'] for q in st if q['name'].startswith('ry_igemm_ldsdma<') and q['name'][:-1].split(',')[5] == 'true']
    exact = [q['layer'] for q in st if q['name'].startswith('ry_igemm_ldsdma<') and q['name'][:-1].split(',')[5] == 'false']
    assert 'encoder/c1' in split and 'decoder/c6' in split and 'encoder/c2' in exact, (split, exact)
    assert [q['name'] for q in st if q['layer'] == 'decoder/c7'] == ['ry_sr_last<false>']
    assert not numpy.array_equal(y3, y32)
    assert rel_max(y3, ref) < 2e-5, rel_max(y3, ref)
    monkeypatch.setenv('RY_X3_MINM', '1')             # every implicit-GEMM layer on the split path (mixed-format copies gone)
    net.set_dtype('bf16x3')
    assert rel_max(net.forward(x), ref) < 2e-5
    net.set_dtype('f32')
    assert numpy.array_equal(net.forward(x), y32)     # the exact path comes back bit for bit
    # the SuperResolution.convert wrapper (pad 'minimum' / log / drop bin ... exp / edge / crop fused around the predictor) in both modes
    sp = numpy.exp(numpy.random.default_rng(23).normal(-6.0, 1.5, (11, 17))).astype('f4')
    c32 = net.convert(sp)
    net.set_dtype('bf16x3')
    c3 = net.convert(sp)
    assert c3.shape == (11, 17) and not numpy.array_equal(c3, c32)
    assert float(numpy.abs(numpy.log(c3) - numpy.log(c32)).max() / numpy.abs(numpy.log(c32)).max()) < 2e-5
    net.close()


def test_stage2_identical_padding_rows_emu(emu_ctx, monkeypatch):
    """The encoder's rows behind the real frames that equal the row above them are copied, not computed (RY_S2_HOLE): split-free implicit-GEMM layers
    leave whole tile rows out of their grids; bit-identical to computing them, also around a discard and for two windows per call."""
    import ctypes
    reread = lambda: emu_ctx.reload_env()
    d = NetDesc(2, 1, 1, 64, 3)
    P = synthetic_params(d, 451, bias_std=0.05)
    sp = numpy.exp(numpy.random.default_rng(72).normal(-6.0, 1.5, (70, 65))).astype('f4')
    try:
        monkeypatch.setenv('RY_PLAN', '1:3:1:1,2:4:1:1'); reread()          # encoder c1 / c2 without split-K (as at full size), 64- and 32-row tiles
        net = engine.Net(emu_ctx, d, flatten_params(d, P), width=64)
        y1 = net.convert(sp)
        reps = [q for q in net.profile(1, 70, 1, window=True) if q['name'] == 'ry_rep_rows']
        assert {q['layer'] for q in reps} == {'encoder/c1', 'encoder/c2'}, reps
        assert float(numpy.abs(y1 / unet.stage2_convert(sp, P, 3) - 1).max()) < cases.TOL
        part = net.convert(sp, discard=(20, 30))
        two = net.convert(numpy.stack([sp, sp[::-1]]))
        monkeypatch.setenv('RY_S2_HOLE', '0'); reread(); net.set_dtype('f32')
        assert not [q for q in net.profile(1, 70, 1, window=True) if q['name'] == 'ry_rep_rows']
        y0 = net.convert(sp)
        assert numpy.array_equal(y0, y1)
        assert numpy.array_equal(part[20:40], y0[20:40])
        assert numpy.array_equal(net.convert(numpy.stack([sp, sp[::-1]])), two)
        net.close()
    finally:
        monkeypatch.delenv('RY_S2_HOLE', raising=False); monkeypatch.delenv('RY_PLAN', raising=False); reread()


def test_stage2_winograd_layers_emu(emu_ctx, monkeypatch):
    """Round 6: the k4 s2 p1 layers with enough rows run in Winograd F(2x2, 2x2) form (PATH_WINO) inside the predictor -- convolutions over four parity
    planes, sub-pixel deconvolutions over a two-source skip concat -- in the convert wrapper with the copied padding rows, the dead-row crop, a discard and
    two windows per call (bit-identical among themselves); RY_WINOGRAD=0 is the direct form of the same predictor (1e-5 apart, both on the oracle)."""
    d = NetDesc(2, 1, 1, 64, 3)
    P = synthetic_params(d, 451, bias_std=0.05)
    sp = numpy.exp(numpy.random.default_rng(72).normal(-6.0, 1.5, (70, 65))).astype('f4')
    ref = unet.stage2_convert(sp, P, 3)
    wino = lambda net: {q['layer'] for q in net.profile(1, 70, 1, window=True) if q['name'].startswith('ry_wino_ldsdma<')}
    try:
        monkeypatch.setenv('RY_WINO_MINM', '1'); emu_ctx.reload_env()
        net = engine.Net(emu_ctx, d, flatten_params(d, P), width=64)
        y = net.convert(sp)
        assert wino(net) == {'encoder/c1', 'encoder/c2', 'decoder/c5', 'decoder/c6'}
        st = net.profile(1, 70, 1, window=True)
        assert all(abs(q['flops_exec'] - (q['flops'] * 9 / 16 if q['name'].startswith('ry_wino') else q['flops'])) <= 1e-6 * q['flops'] for q in st), st
        assert float(numpy.abs(y / ref - 1).max()) < cases.TOL
        monkeypatch.setenv('RY_WINOGRAD', '0'); emu_ctx.reload_env(); net.set_dtype('f32')
        assert not wino(net)
        yd = net.convert(sp)
        assert 0 < float(numpy.abs(y / yd - 1).max()) < 2e-5 and float(numpy.abs(yd / ref - 1).max()) < cases.TOL
        # both workgroup shapes, split-free so that the encoder's identical padding rows are left out of the grids (ry_rep_rows)
        for spec in ('1:1:2:2,2:1:1:1,13:1:1:1,14:1:2:1', '1:2:2:1,2:2:1:1,13:2:1:2,14:2:2:1'):      # (encoder c1 of the first: copied rows AND an external split -- reduce node, then copy node)
            monkeypatch.setenv('RY_WINOGRAD', '1'); monkeypatch.setenv('RY_WINO', spec); monkeypatch.delenv('RY_S2_HOLE', raising=False)
            emu_ctx.reload_env(); net.set_dtype('f32')
            names = {q['layer']: q['name'] for q in net.profile(1, 70, 1, window=True) if q['name'].startswith('ry_wino')}
            cfg = spec.split(',')[0].split(':')[1]
            assert names['encoder/c1'] == ('ry_wino_ldsdma<2,2,1,2>' if cfg == '1' else 'ry_wino_ldsdma<4,2,2,2>') and names['decoder/c6'][-3:] == ',1>', names
            if cfg == '1':       # (8-row tiles: rows 40 .. 55 of encoder c1 are whole tile rows inside the stretch of identical rows; the 16-row tiles of the other spec have none)
                assert {q['layer'] for q in net.profile(1, 70, 1, window=True) if q['name'] == 'ry_rep_rows'} >= {'encoder/c1'}
            y1 = net.convert(sp)
            assert float(numpy.abs(y1 / ref - 1).max()) < cases.TOL
            part = net.convert(sp, discard=(20, 30))
            two = net.convert(numpy.stack([sp, sp[::-1]]))
            assert numpy.array_equal(part[20:40], y1[20:40]) and float(numpy.abs(two[1] / unet.stage2_convert(sp[::-1], P, 3) - 1).max()) < cases.TOL
            monkeypatch.setenv('RY_S2_HOLE', '0'); monkeypatch.setenv('RY_S2_CROP', '0'); emu_ctx.reload_env(); net.set_dtype('f32')
            assert not [q for q in net.profile(1, 70, 1, window=True) if q['name'] == 'ry_rep_rows']
            assert numpy.array_equal(net.convert(sp), y1)                    # computing the copied rows and the cropped rows changes nothing
            assert numpy.array_equal(net.convert(numpy.stack([sp, sp[::-1]])), two)       # (two windows per call: the planner's splits of the other layers may differ from one window's)
            monkeypatch.delenv('RY_S2_HOLE'); monkeypatch.delenv('RY_S2_CROP')
        # a forced plan that does not exist is refused
        monkeypatch.setenv('RY_WINO', '1:1:4:1'); emu_ctx.reload_env()
        with pytest.raises(RuntimeError, match='RY_WINO'):
            net.set_dtype('f32'); net.convert(sp)
        net.close()
    finally:
        for k in ('RY_WINO_MINM', 'RY_WINOGRAD', 'RY_WINO', 'RY_S2_HOLE', 'RY_S2_CROP'):
            monkeypatch.delenv(k, raising=False)
        emu_ctx.reload_env()


def test_stage2_output_stationary_layers_emu(emu_ctx, monkeypatch):
    """Round 5: layers with few rows per phase and the ry_c2d_os filter layout run output-stationary (PATH_OS2D: one node, no slabs) inside the
    predictor -- k4 s2 convolutions, sub-pixel deconvolutions over a two-source skip concat -- next to implicit-GEMM neighbours, in the
    convert wrapper with the dead-row crop / a discard (bit-identical kept rows), and as producers of split-bf16 copies in 'bf16x3' mode."""
    import ctypes
    reread = lambda: emu_ctx.reload_env()
    d = NetDesc(2, 1, 1, 64, 3)
    P = synthetic_params(d, 451, bias_std=0.05)
    x = numpy.random.default_rng(71).normal(size=(1, 16, 16)).astype('f4')
    ref = cases.oracle_forward(d, P, x)
    sp = numpy.exp(numpy.random.default_rng(72).normal(-6.0, 1.5, (11, 17))).astype('f4')
    try:
        monkeypatch.setenv('RY_OS2_MINW', '1'); reread()          # every eligible layer gets the layout (the product keeps it for filters >= 8 MB)
        net = engine.Net(emu_ctx, d, flatten_params(d, P), width=16)
        y = net.forward(x)
        assert rel_max(y, ref) < cases.TOL
        os_layers = [q['layer'] for q in net.profile(1, 16, 1) if q['name'].startswith('ry_c2d_os<')]
        assert {'decoder/c1', 'decoder/c2', 'decoder/c3', 'decoder/c5'} <= set(os_layers), os_layers
        assert not any(q['layer'] in os_layers for q in net.profile(1, 16, 1) if q['name'].startswith('ry_splitk_reduce'))
        monkeypatch.setenv('RY_OS2_MAXCOST', '0'); reread(); net.set_dtype('f32')        # the same predictor on the implicit GEMM: other summation order only
        assert not [q for q in net.profile(1, 16, 1) if q['name'].startswith('ry_c2d_os<')]
        y_ig = net.forward(x)
        assert rel_max(y, y_ig) < 1e-5 and rel_max(y_ig, ref) < cases.TOL
        monkeypatch.setenv('RY_OS2', '14:0,13:1:2:8:2'); monkeypatch.delenv('RY_OS2_MAXCOST'); reread(); net.set_dtype('f32')   # forced per layer: decoder c6 off, decoder c5 on another slice
        names = {q['layer']: q['name'] for q in net.profile(1, 16, 1)}
        assert names['decoder/c5'].startswith('ry_c2d_os<1,2,8,2,') and not names['decoder/c6'].startswith('ry_c2d_os'), names
        assert rel_max(net.forward(x), ref) < cases.TOL
        monkeypatch.delenv('RY_OS2'); reread(); net.set_dtype('f32')
        assert numpy.array_equal(net.forward(x), y)
        # convert wrapper: dead-row crop and a discard hint around output-stationary layers (they run whole and stop the row-range chain)
        monkeypatch.setenv('RY_S2_CROP', '0'); reread()
        whole = net.convert(sp)
        monkeypatch.setenv('RY_S2_CROP', '2'); reread()
        cropped = net.convert(sp)
        assert numpy.array_equal(whole, cropped)
        assert float(numpy.abs(cropped / unet.stage2_convert(sp, P, 3) - 1).max()) < cases.TOL
        part = net.convert(sp, discard=(4, 3))
        assert numpy.array_equal(part[4:8], whole[4:8]) and not part[:4].any() and not part[8:].any()
        # split-bf16 mode: the output-stationary layers stay exact fp32 and write the [hi | lo] copies their bf16 consumers read
        monkeypatch.setenv('RY_X3_MINM', '64')
        net.set_dtype('bf16x3')
        st = net.profile(1, 16, 1)
        names = {q['layer']: q['name'] for q in st if not q['name'].startswith('ry_splitk')}
        assert names['decoder/c5'].startswith('ry_c2d_os<') and names['decoder/c6'].startswith('ry_igemm_ldsdma<') and names['decoder/c6'][:-1].split(',')[5] == 'true', names
        assert rel_max(net.forward(x), ref) < 2e-5
        net.set_dtype('f32')
        assert numpy.array_equal(net.forward(x), y)
        net.close()
    finally:
        for k in ('RY_OS2_MINW', 'RY_OS2_MAXCOST', 'RY_OS2', 'RY_S2_CROP', 'RY_X3_MINM'):
            monkeypatch.delenv(k, raising=False)
        reread()


def test_no_kernel_reads_what_its_producer_did_not_write_emu(emu_ctx, monkeypatch):
    d = NetDesc(2, 1, 1, 64, 3)
    net = engine.Net(emu_ctx, d, flatten_params(d, synthetic_params(d, 471, bias_std=0.05)), width=16)
    sizes = [(n, numpy.exp(numpy.random.default_rng(90 + n).normal(-6.0, 1.5, (n, 17))).astype('f4')) for n in (11, 23)]
    res = cases.poisoned_converts(emu_ctx, net, sizes, monkeypatch)
    assert all(r[2] == 0 and r[3] == 0 for r in res), res
    net.close()


def test_stage2_dead_row_crop_emu(emu_ctx, monkeypatch):
    """The convert wrapper keeps n_frames rows of a window padded to a multiple of 128: decoder layers may skip the rows that only
    feed the discarded padding (LayerPlan::crop_hi).  Same arithmetic on the rows that are kept: results are bit-identical to the
    uncropped run and stay on the oracle (the GPU suite covers more window lengths)."""
    import ctypes
    d = NetDesc(2, 1, 1, 64, 3)
    P = synthetic_params(d, 433, bias_std=0.05)
    net = engine.Net(emu_ctx, d, flatten_params(d, P), width=16)
    reread = lambda: emu_ctx.reload_env()
    try:
        for n in (11,):
            sp = numpy.exp(numpy.random.default_rng(40 + n).normal(-6.0, 1.5, (n, 17))).astype('f4')
            monkeypatch.setenv('RY_S2_CROP', '0'); reread()
            g_whole = {q['layer']: q['grid'] for q in net.profile(1, n, 1, window=True)}
            whole = net.convert(sp) if n == 11 else None
            monkeypatch.setenv('RY_S2_CROP', '2'); reread()
            g_crop = {q['layer']: q['grid'] for q in net.profile(1, n, 1, window=True)}
            if n == 11:
                cropped = net.convert(sp)
                assert numpy.array_equal(whole, cropped), n
                both = net.convert(numpy.stack([sp, sp[::-1]]))         # two windows in one call: a row prefix of EVERY image
                assert numpy.array_equal(both[0], cropped)
                assert float(numpy.abs(both[1] / unet.stage2_convert(sp[::-1], P, 3) - 1).max()) < cases.TOL
                assert float(numpy.abs(cropped / unet.stage2_convert(sp, P, 3) - 1).max()) < cases.TOL
            fewer = [k for k in g_whole if g_crop[k][0] < g_whole[k][0]]
            assert 'decoder/c6' in fewer and all(k.startswith('decoder/') for k in fewer), (n, fewer)
    finally:
        monkeypatch.delenv('RY_S2_CROP', raising=False)
        reread()
    net.close()


@pytest.mark.parametrize('discard', [(4, 3), (6, 0), (200, 0)])
def test_stage2_discarded_frames_are_not_computed_emu(emu_ctx, discard):
    """`ry_sr_convert_rows`: a caller that throws away the first / last frames of a window (ConvertStream.process picks the middle of what it
    converted) says so; the decoder then runs on the row RANGE the kept rows depend on.  Kept rows: bit-identical to the full call;
    discarded rows: zeros; a discard that leaves nothing is ignored; one and two windows per call."""
    d = NetDesc(2, 1, 1, 64, 3)
    P = synthetic_params(d, 435, bias_std=0.05)
    net = engine.Net(emu_ctx, d, flatten_params(d, P), width=16)
    n = 23
    sp = numpy.exp(numpy.random.default_rng(61).normal(-6.0, 1.5, (2, n, 17))).astype('f4')
    full = net.convert(sp[0])
    part = net.convert(sp[0], discard=discard)
    k0 = discard[0] if discard[0] < n else 0
    k1 = n - discard[1] if n - discard[1] > k0 else n
    assert numpy.array_equal(part[k0:k1], full[k0:k1])
    assert not part[:k0].any() and not part[k1:].any()
    if discard == (4, 3):
        both = net.convert(sp, discard=discard)                         # two windows per call: a row range of EVERY image
        assert numpy.array_equal(both[0], part) and not both[1][:k0].any() and not both[1][k1:].any()
        assert numpy.array_equal(net.convert(sp[0]), full)              # ... and the full call afterwards is the full call
    net.close()


def test_autotuned_plans_stay_correct_emu(emu_ctx, monkeypatch):
    """RY_AUTOTUNE=1 (opt-in): candidate launch plans of every implicit-GEMM layer are run on the device when a plan is built and
    the fastest replaces the planner's pick.  The emulator has no clock, so the `pick` field of RY_AUTOTUNE forces a non-default candidate per
    layer (another tile / K-group / split-K combination, slabs re-allocated): results must not move beyond summation order,
    in the exact fp32 mode and in the split-bf16 mode."""
    import ctypes
    d = NetDesc(2, 1, 1, 64, 2)
    P = synthetic_params(d, 430, bias_std=0.05)
    net = engine.Net(emu_ctx, d, flatten_params(d, P), width=16)
    x = numpy.random.default_rng(31).normal(size=(1, 8, 16)).astype('f4')
    ref = cases.oracle_forward(d, P, x)
    reread = lambda: emu_ctx.reload_env()
    monkeypatch.setenv('RY_X3_MINM', '1')
    base = {}
    for mode in ('f32', 'bf16x3'):
        net.set_dtype(mode)
        base[mode] = [(q['layer'], q['name'], q['grid']) for q in net.profile(1, 8, 1)]
    try:
        for pick in (0, 2):
            monkeypatch.setenv('RY_AUTOTUNE', '1:1:3:%d' % pick); reread()                # on : 1 timed round : 3 candidates : take candidate `pick`
            for mode in (('f32',) if pick == 0 else ('f32', 'bf16x3')):
                net.set_dtype(mode)
                if pick != 0 and mode == 'f32':                         # candidate 0 is the planner's pick (covered by every other test); the split-bf16
                    assert rel_max(net.forward(x), ref) < cases.TOL, (pick, mode)   # candidates are checked on the GPU (scripts/gpu_autotune.py)
                plan = [(q['layer'], q['name'], q['grid']) for q in net.profile(1, 8, 1)]
                assert (plan == base[mode]) == (pick == 0), (pick, mode, plan)      # candidate 0 is the planner's pick
    finally:
        monkeypatch.delenv('RY_AUTOTUNE', raising=False)
        reread()                                                       # the session-wide context goes back to the defaults
    net.set_dtype('f32')
    net.close()
