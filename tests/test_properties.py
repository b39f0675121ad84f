"""Property tests (hypothesis) of the host-side rules the hot path rests on -- the pad rule both `convert()` wrappers use, the K-list /
parameter count shared by the Python and the native side, the legality of every launch plan the stage-2 planner can return, the
power-domain thresholds of the device gate, the linear structure of mc2sp that lets `decode_spectrogram` run as one matmul
(/root/reference/realtime_voice_conversion/yukarin_wrapper/voice_changer.py:33-41), the shared-memory transport -- and of two operators
on the emulator over random small shapes.  CPU only."""
import ctypes

import numpy
import pytest

pytest.importorskip('hypothesis')          # property tests need the `hypothesis` package (in this image's wheelhouse; README.md lists it): skip, not a collection error, without it
from hypothesis import HealthCheck, given, settings  # noqa: E402
from hypothesis import strategies as st  # noqa: E402

from conftest import bn_params, rel_max
from oracle import mc2sp as omc
from oracle import ops_numpy as ops
from realtime_yukarin_amd import _lib, gate, netspec, sptk
from realtime_yukarin_amd.netspec import NetDesc
from realtime_yukarin_amd.transport import FeatureQueue

FAST = settings(deadline=None, max_examples=60, suppress_health_check=[HealthCheck.function_scoped_fixture])
SLOW = settings(deadline=None, max_examples=12, suppress_health_check=[HealthCheck.function_scoped_fixture])


@FAST
@given(st.integers(1, 5000))
def test_pad_rule(n):
    """pad = 128 - n % 128: never zero (a whole extra block when n is a multiple), padded length a multiple of 128 (seven halvings)."""
    p = netspec.pad_frames(n)
    assert 1 <= p <= 128 and (n + p) % 128 == 0 and (p == 128) == (n % 128 == 0)


@FAST
@given(st.sampled_from([1, 2]), st.integers(1, 600), st.integers(1, 40), st.sampled_from([1, 3, 8, 16, 64]), st.integers(0, 8), st.booleans())
def test_parameter_count_agrees_between_python_and_the_library(emu_ctx, ndim, in_ch, out_ch, base, e, glu):
    if ndim == 2:
        in_ch = out_ch = 1; glu = False
    d = NetDesc(ndim, in_ch, out_ch, base, e, glu=glu)
    c = _lib.RyNetDesc(d.ndim, d.in_ch, d.out_ch, d.base, d.extensive_layers, 512 if ndim == 2 else 1, 2e-5, 0.2, int(d.glu))
    assert int(emu_ctx.lib.dll.ry_net_param_count(ctypes.byref(c))) == netspec.param_count(d)
    keys = [k for k, _ in netspec.param_list(d)]
    assert len(keys) == len(set(keys)) == 4 + 14 * 6                       # the K-list of SURVEY.md 8(c) item 3, whatever the sizes


@FAST
@given(st.integers(1, 200000), st.sampled_from([64, 128, 192, 256, 512, 1024]), st.sampled_from([1, 4]), st.integers(1, 512))
def test_every_plan_the_planner_returns_is_launchable(emu_ctx, M, cout, nphases, nk):
    t, s, g, e = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
    emu_ctx.lib.check(emu_ctx.lib.dll.ry_debug_plan_igemm(M, cout, nphases, nk, ctypes.byref(t), ctypes.byref(s), ctypes.byref(g), ctypes.byref(e)))
    bm, bn = {1: (128, 128), 3: (64, 128), 4: (32, 128), 5: (128, 64), 6: (96, 128)}[t.value]
    assert cout % bn == 0 and 1 <= s.value <= nk and g.value in (1, 2) and s.value * g.value <= max(nk, 1) and e.value > 0
    assert s.value <= 128 and (bm >= 64 or M <= 32 or cout % 128 != 0)


@FAST
@given(st.floats(5.0, 130.0))
def test_gate_thresholds_are_the_host_predicate(thr):
    """p_effective is the smallest float32 power the host formula 10 log10(max(1e-10, p)) > -thr accepts; p_all the clamp point."""
    p_eff, p_all = gate.thresholds(thr)
    db = lambda p: 10.0 * numpy.log10(numpy.maximum(1e-10, numpy.float32(p)))
    assert db(p_eff) > -thr
    below = numpy.nextafter(numpy.float32(p_eff), numpy.float32(0))
    assert below <= 0 or not (db(below) > -thr) or numpy.float32(p_eff) <= numpy.float32(1e-10)
    assert p_all >= p_eff


@FAST
@given(st.integers(0, 2 ** 31 - 1), st.sampled_from([(0.41, 1024), (0.466, 1024), (0.544, 256)]))
def test_mc2sp_is_the_exponential_of_a_linear_map(seed, cfg):
    """exp(Re rfft(sym(freqt(a + b)))) = mc2sp(a) * mc2sp(b): the property that lets decode_spectrogram be ONE matmul + exp on the device."""
    alpha, fftlen = cfg
    rng = numpy.random.default_rng(seed)
    a, b = rng.normal(size=(2, 3, 9)) * numpy.array([2, 1, .5, .5, .3, .3, .2, .2, .2])
    lhs = omc.mc2sp(a + b, alpha, fftlen)
    assert float(numpy.abs(lhs / (omc.mc2sp(a, alpha, fftlen) * omc.mc2sp(b, alpha, fftlen)) - 1).max()) < 1e-12
    M = sptk.mc2sp_matrix(8, alpha, fftlen)
    assert float(numpy.abs(numpy.exp((a + b) @ M) / lhs - 1).max()) < 1e-11
    assert float(numpy.abs(omc.mc2sp(numpy.zeros((1, 9)), alpha, fftlen) - 1).max()) == 0.0


arrays = st.builds(lambda shape, dt, seed: (numpy.random.default_rng(seed).normal(size=shape) * 100).astype(dt),
                   st.lists(st.integers(0, 17), min_size=0, max_size=3).map(tuple), st.sampled_from(['f4', 'f8', 'i4', 'u1', '?']), st.integers(0, 1000))
items = st.recursive(st.one_of(arrays, st.integers(-5, 5), st.text(max_size=5), st.none()),
                     lambda ch: st.one_of(st.lists(ch, max_size=3), st.dictionaries(st.text(max_size=3), ch, max_size=3)), max_leaves=8)


def same(a, b):
    if isinstance(a, numpy.ndarray):
        return isinstance(b, numpy.ndarray) and a.dtype == b.dtype and a.shape == b.shape and numpy.array_equal(a, b)
    if isinstance(a, (list, tuple)):
        return type(a) is type(b) and len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    if isinstance(a, dict):
        return isinstance(b, dict) and a.keys() == b.keys() and all(same(a[k], b[k]) for k in a)
    return a == b and type(a) is type(b)


@pytest.fixture(scope='module')
def ring():
    q = FeatureQueue(slots=3, slot_bytes=1 << 20)
    yield q
    q.close()


@FAST
@given(st.lists(items, min_size=1, max_size=5))
def test_feature_queue_returns_what_was_put_in_order(ring, objs):
    """Any picklable item, arrays of any dtype / shape (empty and 0-d included, non-contiguous ones too), FIFO over ring laps."""
    out = []
    for o in objs:
        if isinstance(o, numpy.ndarray) and o.ndim == 2 and o.shape[1] > 1:
            o = o[:, ::2]                                               # a view: not contiguous
        ring.put(o)
        out.append((o, ring.get(timeout=5)))
    assert ring.empty() and all(same(a, b) for a, b in out)


@SLOW
@given(st.integers(1, 2), st.integers(1, 40), st.sampled_from([3, 8, 9, 20, 33]), st.sampled_from([4, 10, 64, 70]), st.sampled_from([(4, 2, 1, False), (4, 2, 1, True), (3, 1, 1, False), (1, 1, 0, False)]),
       st.sampled_from([None, 'relu', 'lrelu']), st.integers(0, 3), st.integers(0, 10 ** 6))
def test_conv1d_operator_on_random_small_shapes_emu(emu_ctx, B, L, cin, cout, geom, act, splits, seed):
    k, s, p, tr = geom
    if not tr and L + 2 * p < k:
        L = k
    rng = numpy.random.default_rng(seed)
    x = rng.normal(size=(B, L, cin)).astype('f4')
    W = (rng.normal(size=(cin, cout, k) if tr else (cout, cin, k)) * 0.1).astype('f4')
    b = rng.normal(size=cout).astype('f4')
    bn = bn_params(rng, cout)
    y = emu_ctx.conv1d(x, W, b, bn, stride=s, pad=p, transposed=tr, act=act, splits=min(splits, cin))
    xn = x.transpose(0, 2, 1)
    r = ops.deconv_nd(xn, W, b, stride=s, pad=p) if tr else ops.conv_nd(xn, W, b, stride=s, pad=p)
    r = ops.apply_act(ops.batch_norm_inference(r, *bn), act).transpose(0, 2, 1)
    assert y.shape == r.shape and rel_max(y, r) < 1e-4
