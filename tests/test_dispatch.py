"""The multi-GPU dispatcher of the chained path (realtime_yukarin_amd/dispatch.py): window k -> GPU k mod G, one process per GPU, results
released in `Item.index` order (/root/reference/run.py:171-183), windows built centrally by the stream's `fetch`
(/root/reference/realtime_voice_conversion/stream/base_stream.py:32-79).

CPU suite: two emulator workers under injected jitter -- worker 0 is slow, so windows finish OUT of order -- must release in order and
return the bits the single-worker run returns; a worker that dies is reported, not waited for.  `-m gpu`: G = 1 through the same code
on the real GPU with the weights arriving by a (one-rank) RCCL broadcast, against the composed oracle."""
import importlib
import pickle
import threading
import time
from pathlib import Path

import numpy
import pytest

import test_shims_e2e as e2e
from realtime_yukarin_amd import compat, dispatch, engine

compat.install()
ROOT = Path(__file__).resolve().parent.parent
REF = Path('/root/reference')
KEYS = ('f0', 'ap', 'sp', 'voiced', 'mc')


from dispatch_hooks import (dying_hook, dying_mid_stream_hook, emu_hook, failing_window_hook, jitter_hook, long_stall_hook,
                            stall_hook)      # top-level functions of a LIGHT module: every spawned worker imports the module of its hook


def windows(n_frames, count):
    from yukarin import AcousticFeature, Wave
    out = []
    for i in range(count):
        wave, feat = e2e.make_window(n_frames, 700 + i)

        class_feat = AcousticFeature(**{k: v.copy() for k, v in feat.items()})
        class_feat.wave = Wave(wave=wave, sampling_rate=e2e.FS)
        out.append((wave, feat, class_feat))
    return out


def run(ac, sr, devices, wins, hook, comm='host', pad=7, mp_context='spawn'):
    pick = (pad, -pad, ('f0', 'ap', 'sp', 'voiced', 'mc')) if pad > 0 else None
    with dispatch.ChunkDispatcher(ac, sr, devices, threshold=60, comm=comm, worker_hook=hook, mp_context=mp_context, start_timeout=900) as d:
        got, workers = [], []
        for i, (_, _, f) in enumerate(wins):
            workers.append(d.submit(100 + i, f, discard=(pad, pad), pick=pick))
            got += d.collect()
        got += d.drain(timeout=900)
        return got, workers, d.max_out_of_order


def same(a, b):
    return all(numpy.array_equal(getattr(a, k), getattr(b, k)) for k in KEYS)


def test_two_workers_out_of_order_completion_in_order_release_same_bits(tmp_path):
    e2e.write_models(tmp_path, 'SYN-8')
    ac, sr = e2e.build_converters(tmp_path)
    wins = windows(24, 4)
    one, w1, _ = run(ac, sr, [0], wins, emu_hook)
    two, w2, ooo = run(ac, sr, [0, 0], wins, emu_hook)
    assert w1 == [0] * 4 and w2 == [0, 1, 0, 1]                         # window k -> worker k mod G
    assert [i for i, _ in one] == [i for i, _ in two] == list(range(100, 104))   # released in submission (= Item.index) order
    assert ooo >= 1, 'the fast worker must have finished windows ahead of the slow one (otherwise the test shows nothing)'
    for (_, a), (_, b) in zip(one, two):
        assert a.sp.shape == (24 - 14, 513) and same(a, b)                       # only the kept frames travel back; bit for bit
    # ... and they are what the in-process window call returns
    from realtime_yukarin_amd import _lib, build
    from realtime_yukarin_amd.voice_changer import VoiceChanger
    ctx = engine.Context(0, _lib.Ry355Lib(build.build_emu()))
    real = engine.get_context
    engine.get_context = lambda device=0, lib=None: ctx
    try:
        vc = VoiceChanger(acoustic_converter=ac, super_resolution=sr, threshold=60)
        for (_, out), (_, _, f) in zip(two, wins):
            ref = vc.convert_from_acoustic_feature(f, discard=(7, 7)).pick(7, -7, keys=list(KEYS))
            assert same(out, ref)
        vc.close(); ac.close(); sr.close()
    finally:
        engine.get_context = real


def test_a_dead_worker_is_reported_not_waited_for(tmp_path):
    e2e.write_models(tmp_path, 'SYN-8')
    ac, sr = e2e.build_converters(tmp_path)
    t0 = time.time()
    with pytest.raises(RuntimeError, match='cannot see its GPU|exited with code'):
        dispatch.ChunkDispatcher(ac, sr, [0, 0], comm='host', worker_hook=dying_hook, start_timeout=600)
    assert time.time() - t0 < 300


def test_a_worker_that_fails_mid_stream_is_reported(tmp_path):
    """An exception inside a worker after start-up (the reference's loop has no try / except either, convert_worker.py:45-59: the worker
    dies) reaches the caller of `collect` as a RuntimeError carrying the worker's traceback; the dispatcher closes itself."""
    e2e.write_models(tmp_path, 'SYN-8')
    ac, sr = e2e.build_converters(tmp_path)
    wins = windows(24, 3)
    d = dispatch.ChunkDispatcher(ac, sr, [0], threshold=60, comm='host', worker_hook=failing_window_hook, start_timeout=600)
    with pytest.raises(RuntimeError, match='device lost'):
        for i, (_, _, f) in enumerate(wins):
            d.submit(i, f)
        d.drain(timeout=300)
    assert d.closed
    with pytest.raises(RuntimeError, match='closed'):
        d.submit(9, wins[0][2])


def test_a_stalled_worker_with_a_backlog_behind_it_does_not_deadlock(tmp_path):
    """Round-3 advisor: with bounded rings everywhere, a caller that only submits while one GPU stalls used to end with the caller stuck in
    `put` on the stalled worker's ring and every worker stuck in `put` on the return ring (nobody taking results).  `submit` now keeps
    taking finished windows while it waits for a free slot, and every worker has a return ring of its own.  Null workers (no arithmetic):
    two-slot rings, worker 0 sleeps on its second window, forty windows submitted back to back."""
    e2e.write_models(tmp_path, 'SYN-8')
    ac, sr = e2e.build_converters(tmp_path)
    wins = windows(24, 3)
    t0 = time.time()
    with dispatch.ChunkDispatcher(ac, sr, [0, 0], comm='host', null_workers=True, worker_hook=stall_hook, slots=2, start_timeout=600) as d:
        got = []
        for i in range(40):
            d.submit(i, wins[i % 3][2], discard=(7, 7), pick=(7, -7, KEYS))
        got += d.drain(timeout=120)
    assert [i for i, _ in got] == list(range(40)) and time.time() - t0 < 60
    assert d.max_out_of_order >= 2                                   # worker 1 ran ahead while worker 0 slept
    for _, f in got:
        assert f.sp.shape == (10, 513) and f.ap.shape == (10, 513)


def test_eight_workers_jitter_small_rings_in_order_release(tmp_path):
    """Round-5 verdict, item 7: the hand-out at the node's real width.  G = 8 null workers (no arithmetic; the rings, the round-robin hand-out and the
    in-order release are the product's), two-slot rings, every worker at its own changing pace with a long stall somewhere: 240 windows come back in
    submission order, every worker got every eighth window, completions ran far ahead of the release point, nothing deadlocks."""
    e2e.write_models(tmp_path, 'SYN-8')
    ac, sr = e2e.build_converters(tmp_path)
    wins = windows(24, 3)
    t0 = time.time()
    with dispatch.ChunkDispatcher(ac, sr, [0] * 8, comm='host', null_workers=True, worker_hook=jitter_hook, slots=2, start_timeout=600) as d:
        got, workers = [], []
        for i in range(240):
            workers.append(d.submit(1000 + i, wins[i % 3][2], discard=(7, 7), pick=(7, -7, KEYS)))
            got += d.collect()
        got += d.drain(timeout=180)
        ooo = d.max_out_of_order
    assert workers == [i % 8 for i in range(240)]
    assert [i for i, _ in got] == list(range(1000, 1240)) and time.time() - t0 < 120
    assert ooo >= 4, ooo                                              # several workers finished windows ahead of the one being waited for
    assert all(f.sp.shape == (10, 513) and f.ap.shape == (10, 513) for _, f in got)


def test_eight_emulator_workers_return_the_single_worker_bits(tmp_path):
    """... and with the arithmetic: eight emulator worker processes (SYN-8) against one, bit for bit, in order."""
    e2e.write_models(tmp_path, 'SYN-8')
    ac, sr = e2e.build_converters(tmp_path)
    wins = windows(24, 8)
    one, w1, _ = run(ac, sr, [0], wins, emu_hook)
    eight, w8, _ = run(ac, sr, [0] * 8, wins, emu_hook)
    assert w8 == list(range(8)) and [i for i, _ in eight] == [i for i, _ in one] == list(range(100, 108))
    for (_, a), (_, b) in zip(one, eight):
        assert same(a, b)


def test_one_of_eight_workers_dies_mid_stream(tmp_path):
    """Worker 5 of eight is lost after start-up with windows queued on every ring: the caller gets the worker's error (not a hang), the dispatcher closes
    itself and the other seven processes are gone afterwards."""
    e2e.write_models(tmp_path, 'SYN-8')
    ac, sr = e2e.build_converters(tmp_path)
    wins = windows(24, 3)
    t0 = time.time()
    d = dispatch.ChunkDispatcher(ac, sr, [0] * 8, comm='host', null_workers=True, worker_hook=dying_mid_stream_hook, slots=2, start_timeout=600)
    with pytest.raises(RuntimeError, match='device lost'):
        for i in range(64):
            d.submit(i, wins[i % 3][2], discard=(7, 7), pick=(7, -7, KEYS))
        d.drain(timeout=120)
    assert d.closed and time.time() - t0 < 120
    assert not any(p.is_alive() for p in d._procs)


def test_close_keeps_its_deadline_with_a_stalled_worker_among_eight(tmp_path):
    """close() with work in flight and one worker asleep for 8 s: one deadline for all workers (20 s), the sleeper is joined when it wakes, nobody is left."""
    e2e.write_models(tmp_path, 'SYN-8')
    ac, sr = e2e.build_converters(tmp_path)
    wins = windows(24, 3)
    d = dispatch.ChunkDispatcher(ac, sr, [0] * 8, comm='host', null_workers=True, worker_hook=long_stall_hook, slots=2, start_timeout=600)
    for i in range(16):
        d.submit(i, wins[i % 3][2], discard=(7, 7), pick=(7, -7, KEYS))
    t0 = time.time()
    d.close()
    assert d.closed and time.time() - t0 < 30
    assert not any(p.is_alive() for p in d._procs)


def test_lean_windows_and_whole_objects_return_the_same_bits(tmp_path):
    """A window travels without its `ap` block (re-attached by the dispatcher, zeroed on the frames the silence gate cut) and as plain
    arrays without pickle; `lean=False` ships the whole feature object both ways as round 3 did.  Same bits either way, every key, with and
    without a pick -- the windows of `make_window` have a silent and a quiet stretch, so the re-attached `ap` has zero rows."""
    e2e.write_models(tmp_path, 'SYN-8')
    ac, sr = e2e.build_converters(tmp_path)
    wins = windows(64, 3)                                          # long enough for frames whose whole 1024-sample gate window is silent
    res = {}
    for lean in (True, False):
        with dispatch.ChunkDispatcher(ac, sr, [0], threshold=60, comm='host', worker_hook=emu_hook, lean=lean, start_timeout=900) as d:
            assert d.lean == lean
            for i, (_, _, f) in enumerate(wins):
                d.submit(i, f, discard=(7, 7), pick=(7, -7, KEYS))
            d.submit(3, wins[0][2])                                  # no pick: every frame, every key convert_from_acoustic_feature sets
            res[lean] = d.drain(timeout=900)
    for (ia, a), (ib, b) in zip(res[True], res[False]):
        assert ia == ib and same(a, b) and a.ap.dtype == b.ap.dtype == numpy.float32
    full = res[True][3][1]
    eff = e2e.oef.separate_effective_mask(wins[0][0], e2e.FS, 64, 60, 1024, e2e.FRAME_PERIOD)        # the oracle's gate on the same wave
    assert full.sp.shape == (64, 513) and 0 < eff.sum() < 64
    assert not full.ap[~eff].any() and numpy.array_equal(full.ap[eff], wins[0][1]['ap'][eff])


def test_weightless_copies_for_the_broadcast_receivers(tmp_path):
    """What the dispatcher ships to GPUs 1 .. G-1 when the weights travel by RCCL: the config without the host weights."""
    e2e.write_models(tmp_path, 'SYN-8')
    ac, sr = e2e.build_converters(tmp_path)
    la, ls = ac.without_weights(), sr.without_weights()
    assert la._params is None and ls._params is None and ac._params is not None and sr._params is not None
    assert len(pickle.dumps(la)) < len(pickle.dumps(ac)) / 10
    assert la.desc == ac.desc and la.f0_converter is ac.f0_converter and ls.desc == sr.desc
    with pytest.raises(RuntimeError, match='carries no weights'):
        la._get_net()
    with pytest.raises(RuntimeError, match='carries no weights'):
        ls._get_net(513)


@pytest.mark.skipif(not REF.exists(), reason='/root/reference is not mounted here')
def test_drop_in_worker_over_two_gpus_matches_the_single_gpu_mirror(tmp_path, emu_ctx, monkeypatch):
    """`convert_worker_multi_gpu` (the reference's ConvertStream for add / fetch, two emulator workers) against the single-GPU mirror
    worker on the same items: same windows out, in order."""
    for p in (str(ROOT / 'tests' / 'stubs'), str(REF)):
        monkeypatch.syspath_prepend(p)
    vc_mod = importlib.import_module('realtime_voice_conversion.yukarin_wrapper.voice_changer')
    util = importlib.import_module('realtime_voice_conversion.worker.utility')
    from realtime_yukarin_amd import worker
    from realtime_yukarin_amd.transport import FeatureQueue
    monkeypatch.setattr(engine, 'get_context', lambda device=0, lib=None: emu_ctx)
    e2e.write_models(tmp_path, 'SYN-8')
    ac, sr = e2e.build_converters(tmp_path)
    time_length, extra_time, n_items = 0.12, 0.04, 3
    inputs = []
    for i in range(n_items):
        wave, feat = e2e.make_window(24, 900 + i)
        inputs.append(vc_mod.AcousticFeatureWrapper(wave=e2e_wave(wave), **feat))

    def drive(target, extra_kwargs):
        q_in, q_out = FeatureQueue(slots=8, slot_bytes=4 << 20), FeatureQueue(slots=8, slot_bytes=4 << 20)
        lock = threading.Lock(); lock.acquire()
        t = threading.Thread(target=target, args=(ac, sr, time_length, extra_time, 60, q_in, q_out, lock), kwargs=extra_kwargs, daemon=True)
        t.start()
        for i, f in enumerate(inputs):
            q_in.put(util.Item(item=f, index=i))
        q_in.put(None)
        got = [q_out.get(timeout=900) for _ in range(n_items)]
        t.join(timeout=120)
        assert not t.is_alive()
        q_in.close(); q_out.close()
        return got
    try:
        want = drive(worker.convert_worker, {})
        got = drive(dispatch.convert_worker_multi_gpu, dict(devices=[0, 0], comm='host', worker_hook=emu_hook, mp_context='spawn'))
    finally:
        import sys
        for m in [k for k in sys.modules if k.startswith('realtime_voice_conversion')]:  # imported over tests/stubs: not for later tests
            sys.modules.pop(m)
    assert [g.index for g in got] == [w.index for w in want] == list(range(n_items))
    for g, w in zip(got, want):
        assert g.item.sp.shape == w.item.sp.shape == (24, 513)
        for k in ('f0', 'ap', 'sp', 'voiced'):
            assert numpy.array_equal(getattr(g.item, k), getattr(w.item, k)), k


def e2e_wave(wave):
    from yukarin import Wave
    return Wave(wave=wave, sampling_rate=e2e.FS)


@pytest.mark.gpu
def test_dispatcher_with_one_gpu_and_the_rccl_broadcast_gpu(tmp_path, gpu_ctx):
    """G = 1 on the real GPU through the same code: a spawned worker process, the predictors arriving by a one-rank RCCL broadcast
    (`ry_comm_bcast_weights`) and adopted by the shims, windows through the shared-memory rings, against the composed oracle."""
    from oracle import torch_ref
    P1, P2 = e2e.write_models(tmp_path, 'SYN-64')
    ac, sr = e2e.build_converters(tmp_path)
    t1, t2 = torch_ref.TorchUNet(P1), torch_ref.TorchUNet(P2)
    wins = windows(300, 4)
    got, workers, _ = run(ac, sr, [0], wins, None, comm='native', pad=0)
    assert [i for i, _ in got] == [100, 101, 102, 103] and workers == [0, 0, 0, 0]
    for (_, out), (wave, feat, _) in zip(got, wins):
        e2e.check(out, e2e.expected(t1, t2, ac.f0_converter, wave, feat, 300, 60), 300, 'dispatcher G=1 (RCCL broadcast)')


@pytest.mark.gpu
def test_two_worker_processes_on_one_gpu_gpu(tmp_path, gpu_ctx):
    """G = 2 worker PROCESSES on the real hardware -- both on GPU 0, the only one of the box, so the weights go to each by pickle
    (`comm='host'`; RCCL refuses two ranks on one device): two HIP contexts, two window cores, the shared-memory rings, round-robin
    hand-out and in-order release against the composed oracle, and against the single-worker run bit for bit."""
    from oracle import torch_ref
    P1, P2 = e2e.write_models(tmp_path, 'SYN-64')
    ac, sr = e2e.build_converters(tmp_path)
    t1, t2 = torch_ref.TorchUNet(P1), torch_ref.TorchUNet(P2)
    wins = windows(300, 6)
    two, workers, _ = run(ac, sr, [0, 0], wins, None, comm='host', pad=100)
    one, _, _ = run(ac, sr, [0], wins, None, comm='host', pad=100)
    assert workers == [0, 1, 0, 1, 0, 1] and [i for i, _ in two] == list(range(100, 106))
    for (_, a), (_, b), (wave, feat, _) in zip(two, one, wins):
        assert a.sp.shape == (100, 513) and all(numpy.array_equal(getattr(a, k), getattr(b, k)) for k in KEYS)
        exp = e2e.expected(t1, t2, ac.f0_converter, wave, feat, 300, 60)
        assert float(numpy.abs(a.sp.astype(numpy.float64) / exp['sp'][100:200] - 1).max()) < 1e-4           # the kept frames of the window
        assert float(numpy.abs(a.mc - exp['mc'][100:200]).max() / numpy.abs(exp['mc']).max()) < 1e-4
