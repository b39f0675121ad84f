"""The window-call API of the C ABI (`ry_vc_*`, include/ry355.h) on the emulator: tickets, the six-slot ring, the split calls and
their error behaviour -- every misuse returns a negative code with a message (`Ry355Error`), nothing aborts, and the handle stays usable
(the reference's worker loop dies on any exception, convert_worker.py:45-59: errors must be reported, not fatal)."""
import ctypes

import numpy
import pytest

from realtime_yukarin_amd import _lib, engine, gate, sptk, synth
from realtime_yukarin_amd.weights import flatten_params


@pytest.fixture(scope='module')
def core(emu_ctx):
    (d1, P1), (d2, P2) = synth.model_params('SYN-8')
    n1 = engine.Net(emu_ctx, d1, flatten_params(d1, P1))
    n2 = engine.Net(emu_ctx, d2, flatten_params(d2, P2), width=128)
    c = engine.VcCore(n1, n2, sptk.mc2sp_matrix(8, sptk.mcepalpha(16000), 256))
    yield c
    c.close(); n1.close(); n2.close()


def window(n, seed, keep=0.7):
    rng = numpy.random.default_rng(seed)
    x = synth.stage1_input(n, seed=seed)[0]
    eff = rng.random(n) < keep
    return x, eff


def test_tickets_come_back_in_any_order_and_only_once(core):
    (xa, ea), (xb, eb), (xc, ec) = window(20, 1), window(33, 2), window(20, 3)
    ref = [core.convert(x[e], e) for x, e in ((xa, ea), (xb, eb), (xc, ec))]
    ta, tb, tc = core.submit(xa[ea], ea), core.submit(xb[eb], eb), core.submit(xc[ec], ec)
    more = [core.submit(xa[ea], ea) for _ in range(3)]
    with pytest.raises(_lib.Ry355Error, match='ring slots are in flight'):
        core.submit(xa[ea], ea)                                                     # a seventh window: the ring has six slots
    core._pending.pop(ta + 6, None)
    for t in more:
        assert numpy.array_equal(core.wait(t)[1], ref[0][1])
    for t, r in ((tc, ref[2]), (ta, ref[0]), (tb, ref[1])):                          # collected out of order
        mc, sp = core.wait(t)
        assert numpy.array_equal(mc, r[0]) and numpy.array_equal(sp, r[1])
    core._pending[ta] = 20
    with pytest.raises(_lib.Ry355Error, match='not in flight'):
        core.wait(ta)                                                               # a ticket is good for one wait
    mc, sp = core.convert(xa[ea], ea)                                               # ... and the handle is still usable
    assert numpy.array_equal(sp, ref[0][1])


def test_bad_arguments_are_reported(core):
    x, e = window(16, 5)
    rows = numpy.nonzero(e)[0].astype(numpy.int32)
    lib, h = core.lib, core.handle
    ip = ctypes.POINTER(ctypes.c_int)
    t = ctypes.c_int()
    bad = rows.copy(); bad[0] = 99
    xs = numpy.ascontiguousarray(x[e])
    assert lib.dll.ry_vc_submit(h, _lib._fptr(xs), bad.ctypes.data_as(ip), len(rows), 16, 1e-16, ctypes.byref(t)) == -1
    assert b'outside the window' in lib.dll.ry_last_error()
    assert lib.dll.ry_vc_submit(h, _lib._fptr(xs), rows.ctypes.data_as(ip), 17, 16, 1e-16, ctypes.byref(t)) == -1     # more effective frames than frames
    assert lib.dll.ry_vc_submit(h, _lib._fptr(xs), rows.ctypes.data_as(ip), len(rows), 0, 1e-16, ctypes.byref(t)) == -1
    assert lib.dll.ry_vc_submit(h, _lib._fptr(None), rows.ctypes.data_as(ip), len(rows), 16, 1e-16, ctypes.byref(t)) == -1   # null input with rows to convert
    out = numpy.empty((16, core.F), numpy.float32)
    assert lib.dll.ry_vc_wait(h, -3, _lib._fptr(out), _lib._fptr(out)) == -1
    with pytest.raises(_lib.Ry355Error, match='power of two'):
        core.gate(numpy.ones(400, numpy.float32), 80, 1000, 1e-6, 1.0, numpy.zeros((6, 9), numpy.float32))
    with pytest.raises(_lib.Ry355Error, match='power of two'):
        core.submit_wave(numpy.ones(400, numpy.float32), 80, 2048, 1e-6, 1.0, numpy.zeros((6, 9), numpy.float32))
    core._pending.clear()
    mc, sp = core.convert(x[e], e)                                                  # still usable
    assert sp.shape == (16, core.F)


def test_split_calls_follow_the_order_of_the_reference_steps(core):
    x, e = window(24, 7)
    whole = core.convert(x[e], e)
    y1 = core.convert_stage1(x[e])
    assert numpy.array_equal(y1, whole[0][e])
    with pytest.raises(_lib.Ry355Error, match='converted rows on the device'):
        core.stage2_from_mc(numpy.ones(24, bool), 1e-16)                            # asks for 24 rows, stage 1 left fewer
    assert numpy.array_equal(core.stage2_from_mc(e, 1e-16), whole[1])
    mid = core.mid_sp(e, 1e-16)                                                     # the rows are still there after stage 2
    want = numpy.exp(whole[0].astype(numpy.float64) @ sptk.mc2sp_matrix(8, sptk.mcepalpha(16000), 256)) + 1e-16
    assert float(numpy.abs(mid / want - 1).max()) < 2e-5
    sil = numpy.zeros(10, bool)                                                     # an all-silent window needs no stage 1 at all
    assert numpy.array_equal(core.stage2_from_mc(sil, 1e-16), core.convert(numpy.zeros((0, 9), numpy.float32), sil)[1])
    core.convert_stage1(x[e])
    core.wait(core.submit(x[e], e))                                                 # a whole-window call in between: the rows stage 1 left are forgotten
    with pytest.raises(_lib.Ry355Error, match='converted rows on the device'):
        core.stage2_from_mc(e, 1e-16)


def test_stream_generator_depth(core):
    wins = [window(12 + i, 20 + i) for i in range(4)]
    core.reserve(40)
    ref = [core.convert(x[e], e) for x, e in wins]
    for depth in (1, 6):
        got = list(core.convert_stream([(x[e], e) for x, e in wins], depth=depth))
        assert all(numpy.array_equal(g[1], r[1]) and numpy.array_equal(g[0], r[0]) for g, r in zip(got, ref))
    with pytest.raises(ValueError):
        list(core.convert_stream([], depth=7))


def test_batch_call_equals_the_windows_one_by_one(core):
    """`ry_vc_enqueue_device_batch` (several windows of one length per call: stage 2 as one batch): every window of the result is the
    single-window call on that window -- ragged effective counts (stage 1 window by window), equal counts (stage 1 as one batch), an
    all-silent window in the middle, and a bad argument is refused with a message."""
    n = 20
    ws = [window(n, 11), window(n, 12, keep=0.0), window(n, 13, keep=0.4)]
    one = [core.convert(x[e], e) for x, e in ws]
    got = core.convert_batch([(x[e], e) for x, e in ws])
    for (mc, sp), (rmc, rsp) in zip(got, one):
        assert numpy.array_equal(mc, rmc)
        assert float(numpy.abs(sp / rsp - 1).max()) < 1e-5                   # the batch may run under another launch plan: summation order only
    assert numpy.all(got[1][0] == 0)                                          # the all-silent window: zero mc rows (AcousticFeature.silent)
    full = [(x, numpy.ones(n, bool)) for x, _ in ws[:2]]                      # equal effective counts: stage 1 runs as one batch
    got = core.convert_batch(full)
    for (mc, sp), (x, e) in zip(got, full):
        rmc, rsp = core.convert(x, e)
        assert float(numpy.abs(mc - rmc).max()) <= 1e-5 * float(numpy.abs(rmc).max())
        assert float(numpy.abs(sp / rsp - 1).max()) < 1e-5
    core.set_discard(3, 4)                                                    # the frames the caller throws away: not computed, zeros
    try:
        gd = core.convert_batch(full)
        one_d = core.convert(*full[0])
        for (mc, sp), (x, e) in zip(gd, full):
            rmc, rsp = core.convert(x, e)
            assert float(numpy.abs(sp[3:n - 4] / rsp[3:n - 4] - 1).max()) < 1e-5 and not sp[:3].any() and not sp[n - 4:].any()
        assert not one_d[1][:3].any() and not one_d[1][n - 4:].any()
    finally:
        core.set_discard(0, 0)
    assert numpy.array_equal(core.convert(*full[0])[1][3:n - 4], one_d[1][3:n - 4]) and core.convert(*full[0])[1][:3].any()
    with pytest.raises(ValueError, match='one length'):
        core.convert_batch([(ws[0][0][ws[0][1]], ws[0][1]), window(21, 5)])
    ne = (ctypes.c_int * 2)(3, 99)
    with pytest.raises(_lib.Ry355Error, match='bad frame counts'):
        core.lib.check(core.lib.dll.ry_vc_enqueue_device_batch(core.handle, 2, None, None, ne, n, 1e-16, _lib._fptr(1 << 12), _lib._fptr(1 << 12)))
    mc, sp = core.convert(ws[0][0][ws[0][1]], ws[0][1])                      # the handle is still usable
    assert numpy.array_equal(mc, one[0][0]) and numpy.array_equal(sp, one[0][1])


def test_lanes_run_the_same_arithmetic(core):
    """`ry_vc_set_lanes`: ring slot k on its own clone of the predictor pair (shared filters, own streams / plans / activations).  The
    windows of a stream come back bit-identical with 1, 2 and 3 lanes; the lane count cannot change under a window in flight; a clone
    follows the arithmetic mode of the handle it was made from."""
    wins = [window(n, 20 + i) for i, n in enumerate((20, 33, 20, 7, 20))]
    res = {}
    try:
        for lanes in (1, 3):
            core.set_lanes(lanes)
            res[lanes] = list(core.convert_stream([(x[e], e) for x, e in wins], depth=3))
        for lanes in (3,):
            for (mc, sp), (rmc, rsp) in zip(res[lanes], res[1]):
                assert numpy.array_equal(mc, rmc) and numpy.array_equal(sp, rsp)
        t = core.submit(wins[0][0][wins[0][1]], wins[0][1])
        with pytest.raises(_lib.Ry355Error, match='still in flight'):
            core.set_lanes(1)
        core.wait(t)
        with pytest.raises(_lib.Ry355Error, match='lanes must be'):
            core.set_lanes(9)
        # bf16x3 on the caller's stage-2 handle: the clones follow (every slot gives the same answer, different from the fp32 one)
        core.stage2.set_dtype('bf16x3')
        x, e = wins[1]
        a = [core.convert(x[e], e) for _ in range(2)]                           # consecutive slots = different lanes
        assert all(numpy.array_equal(a[0][1], q[1]) for q in a[1:])
        core.stage2.set_dtype('f32')
        b = [core.convert(x[e], e) for _ in range(2)]
        assert all(numpy.array_equal(b[0][1], q[1]) for q in b) and numpy.array_equal(b[0][1], res[1][1][1])
    finally:
        core.stage2.set_dtype('f32')
        core.set_lanes(2)


def test_a_clone_shares_the_filters(emu_ctx):
    (d1, P1), _ = synth.model_params('SYN-8')
    n1 = engine.Net(emu_ctx, d1, flatten_params(d1, P1))
    h = ctypes.c_void_p()
    emu_ctx.lib.check(emu_ctx.lib.dll.ry_net_clone(n1.handle, ctypes.byref(h)))
    x = synth.stage1_input(40, seed=3)[0]
    y = n1.convert(x)
    y2 = numpy.empty_like(y)
    emu_ctx.lib.check(emu_ctx.lib.dll.ry_ac_convert(h, _lib._fptr(x), _lib._fptr(y2), 1, 40, 0))
    assert numpy.array_equal(y, y2)
    n1.close()                                                                  # the clone keeps the filters alive
    emu_ctx.lib.check(emu_ctx.lib.dll.ry_ac_convert(h, _lib._fptr(x), _lib._fptr(y2), 1, 40, 0))
    assert numpy.array_equal(y, y2)
    emu_ctx.lib.dll.ry_net_destroy(h)


def test_a_longer_window_may_arrive_while_others_are_in_flight(emu_ctx):
    """Every ring slot owns its buffers and grows on its own: a stream whose windows get longer (the first fetches of a live stream are
    short) keeps its pipeline; `reserve` sizes all slots ahead of time and refuses only under a window in flight that would have to move."""
    (d1, P1), (d2, P2) = synth.model_params('SYN-8')
    n1 = engine.Net(emu_ctx, d1, flatten_params(d1, P1))
    n2 = engine.Net(emu_ctx, d2, flatten_params(d2, P2), width=128)
    c = engine.VcCore(n1, n2, sptk.mc2sp_matrix(8, sptk.mcepalpha(16000), 256))
    wins = [window(n, 50 + i) for i, n in enumerate((5, 9, 30, 41))]
    ref = [c.convert(x[e], e) for x, e in wins[:1]]                         # the ring starts small
    got = list(c.convert_stream([(x[e], e) for x, e in wins], depth=3))     # 30 and 41 arrive under shorter windows in flight
    c2 = engine.VcCore(n1, n2, sptk.mc2sp_matrix(8, sptk.mcepalpha(16000), 256), lanes=1)
    c2.reserve(64)
    for (mc, sp), (x, e) in zip(got, wins):
        rmc, rsp = c2.convert(x[e], e)
        assert numpy.array_equal(mc, rmc) and numpy.array_equal(sp, rsp)
    assert numpy.array_equal(got[0][1], ref[0][1])
    t = c.submit(wins[0][0][wins[0][1]], wins[0][1])
    with pytest.raises(_lib.Ry355Error, match='still in flight'):
        c.reserve(500)
    c.wait(t)
    c.reserve(500)
    c.warm(9, rounds=1)                                                      # dummy windows through every slot; results unaffected afterwards
    mc, sp = c.convert(wins[1][0][wins[1][1]], wins[1][1])
    assert numpy.array_equal(sp, got[1][1])
    c.close(); c2.close(); n1.close(); n2.close()


def test_closing_a_predictor_takes_its_window_cores_with_it(emu_ctx, monkeypatch):
    """`ry_vc` holds raw pointers to both predictors (their context, streams, lane clones): freeing a predictor under a live core and
    then submitting on it -- or destroying the core afterwards -- would be a use-after-free inside libry355.so.  `Net.close` therefore
    closes the cores built on it first, a closed core refuses politely, and any close order is safe."""
    calls = []
    real_destroy = emu_ctx.lib.dll.ry_vc_destroy
    (d1, P1), (d2, P2) = synth.model_params('SYN-8')
    mtx = sptk.mc2sp_matrix(8, sptk.mcepalpha(16000), 256)
    for order in ('nets first', 'one net only'):
        n1 = engine.Net(emu_ctx, d1, flatten_params(d1, P1))
        n2 = engine.Net(emu_ctx, d2, flatten_params(d2, P2), width=128)
        c = engine.VcCore(n1, n2, mtx)
        x, e = window(12, 7)
        c.convert(x[e], e)
        h = c.handle
        assert c.alive_on(n1, n2)
        if order == 'nets first':
            n1.close()                                  # closes c (while n2 is still alive, so ry_vc_destroy may still dereference both)
            assert c.handle is None and not c.alive_on(n1, n2) and n2.handle is not None
            n2.close(); c.close()                       # closing again is a no-op
        elif order == 'core first':
            c.close(); n1.close(); n2.close()
        else:
            n2.close()
            assert c.handle is None
            n1.close()
        assert not n1._dependents and not n2._dependents
        with pytest.raises(Exception):
            c.convert(x[e], e)                          # a closed core raises in Python; it never reaches the library with a dangling handle


def test_more_than_three_lanes_get_two_ring_slots_each(emu_ctx):
    """`ry_vc_set_lanes(4 .. 8)`: two ring slots per lane (a throughput setting; eight lanes measured within 1 % of two at 300 frames): the ring,
    the lane rotation and the clones are what is checked here -- the same windows come back, in order, with eight in flight."""
    (d1, P1), (d2, P2) = synth.model_params('SYN-8')
    n1 = engine.Net(emu_ctx, d1, flatten_params(d1, P1))
    n2 = engine.Net(emu_ctx, d2, flatten_params(d2, P2), width=128)
    mtx = sptk.mc2sp_matrix(8, sptk.mcepalpha(16000), 256)
    wins = [window(8, 80 + i) for i in range(2)]
    one = engine.VcCore(n1, n2, mtx, lanes=1)
    ref = [one.convert(x[e], e) for x, e in wins]
    one.close()
    c = engine.VcCore(n1, n2, mtx, lanes=4)
    assert c.ring == 8
    tickets = [c.submit(x[e], e) for x, e in wins * 4]                         # eight windows in flight: more than the six slots of the default ring
    with pytest.raises(_lib.Ry355Error, match='all 8 ring slots are in flight'):
        c.submit(wins[0][0][wins[0][1]], wins[0][1])
    got = [c.wait(t) for t in tickets]
    for i, (mc, sp) in enumerate(got):
        assert numpy.array_equal(mc, ref[i % 2][0]) and numpy.array_equal(sp, ref[i % 2][1])
    c.close(); n1.close(); n2.close()
