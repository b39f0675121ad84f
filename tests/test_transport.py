"""Shared-memory feature transport: `FeatureQueue` carries the reference's `Item` objects
(worker/utility.py:6-13) bit-exactly, in order, across processes, with `multiprocessing.Queue`'s calling convention."""
import multiprocessing
import queue

import numpy
import pytest

from realtime_yukarin_amd import compat, transport
from realtime_yukarin_amd.transport import FeatureQueue, echo_worker, measure_round_trip

compat.install()


class Item(object):                      # same shape as realtime_voice_conversion.worker.utility.Item
    def __init__(self, item, index):
        self.item = item
        self.index = index


def make_feature(n=60, seed=0, dtype=numpy.float32):
    from yukarin import AcousticFeature, Wave
    rng = numpy.random.default_rng(seed)
    f = AcousticFeature(f0=rng.random((n, 1)).astype(dtype), ap=rng.random((n, 513)).astype(dtype), sp=rng.random((n, 513)).astype(dtype),
                        mc=rng.normal(size=(n, 9)).astype(dtype), voiced=rng.random((n, 1)) > 0.5)
    f.wave = Wave(wave=rng.normal(size=n * 80).astype(dtype), sampling_rate=16000)
    return f


def same_feature(a, b):
    for k in ('f0', 'ap', 'sp', 'mc', 'voiced'):
        x, y = getattr(a, k), getattr(b, k)
        assert x.dtype == y.dtype and x.shape == y.shape and numpy.array_equal(x, y), k
    assert numpy.array_equal(a.wave.wave, b.wave.wave) and a.wave.sampling_rate == b.wave.sampling_rate


def test_item_round_trip_is_bit_exact_and_owned():
    q = FeatureQueue(slots=2, slot_bytes=1 << 20)
    f = make_feature(dtype=numpy.float64)
    q.put(Item(f, 7))
    assert q.qsize() == 1 and not q.empty()
    got = q.get()
    assert isinstance(got, Item) and got.index == 7 and type(got.item) is type(f)
    same_feature(got.item, f)
    got.item.sp += 1.0                                    # the receiver owns its arrays (voice_changer.py:39 mutates in place)
    q.put(Item(f, 8))
    again = q.get()
    same_feature(again.item, f)
    assert q.empty()
    q.close(); q.close()


def test_order_wraparound_and_queue_exceptions():
    q = FeatureQueue(slots=3, slot_bytes=1 << 16)
    with pytest.raises(queue.Empty):
        q.get_nowait()
    with pytest.raises(queue.Empty):
        q.get(timeout=0.05)
    for i in range(3):
        q.put_nowait(Item(numpy.full(5, i), i))
    assert q.full()
    with pytest.raises(queue.Full):
        q.put_nowait(Item(None, 99))
    for round_ in range(4):                               # ring order survives several laps
        for i in range(3):
            got = q.get()
            assert got.index == i + 3 * round_ and numpy.array_equal(got.item, numpy.full(5, got.index))
            q.put(Item(numpy.full(5, got.index + 3), got.index + 3))
    with pytest.raises(ValueError, match='slot_bytes'):
        q.put(numpy.zeros(1 << 16, numpy.uint8))
    assert q.qsize() == 3                                 # the oversize item took no slot
    q.close()


def test_non_contiguous_empty_and_plain_objects():
    q = FeatureQueue(slots=2, slot_bytes=1 << 20)
    a = numpy.arange(120, dtype=numpy.float32).reshape(10, 12)
    for obj in (a[:, ::3], a.T, a[2:2], numpy.zeros((0, 513), numpy.float32), {'k': [1, 2.5, 'x']}, None):
        q.put(obj)
        got = q.get()
        if isinstance(obj, numpy.ndarray):
            assert got.dtype == obj.dtype and got.shape == obj.shape and numpy.array_equal(got, obj)
        else:
            assert got == obj
    q.close()


@pytest.mark.parametrize('method', ['fork', 'spawn'])
def test_cross_process_echo(method):
    ctx = multiprocessing.get_context(method)
    q_a, q_b = FeatureQueue(slots=4, slot_bytes=2 << 20, ctx=ctx), FeatureQueue(slots=4, slot_bytes=2 << 20, ctx=ctx)
    p = ctx.Process(target=echo_worker, args=(q_a, q_b, 6), daemon=True)
    p.start()
    feats = [make_feature(n=100, seed=s) for s in range(6)]
    for i, f in enumerate(feats[:4]):                     # several in flight
        q_a.put(Item(f, i))
    for i in range(6):
        got = q_b.get(timeout=120)
        assert got.index == i
        same_feature(got.item, feats[i])
        if i + 4 < 6:
            q_a.put(Item(feats[i + 4], i + 4))
    p.join(timeout=30)
    assert p.exitcode == 0
    q_a.close(); q_b.close()


def test_round_trip_harness_runs_for_both_queue_kinds():
    item = Item(make_feature(n=100), 0)
    t_shm = measure_round_trip(lambda: FeatureQueue(slots=4, slot_bytes=4 << 20), item, n=5, warmup=1)
    t_pipe = measure_round_trip(multiprocessing.Queue, item, n=5, warmup=1)
    assert 0 < t_shm < 5 and 0 < t_pipe < 5


def _raw_echo(q_in, q_out, n):
    for _ in range(n):
        m = q_in.get()
        if isinstance(m, transport.Raw):
            q_out.put_arrays(m.tag + 1, [v * 2 for v in m.ints], m.arrays)
        else:
            q_out.put(m)


def test_raw_array_messages_share_a_ring_with_pickled_objects_across_processes():
    """`put_arrays` (the dispatcher's per-window messages: no pickle, a dtype / shape descriptor per array) and ordinary objects in one ring,
    through a spawned child and back: dtypes (float32, bool, float64, int32), shapes (2-D, 1-D, empty), values, order."""
    ctx = multiprocessing.get_context('spawn')
    q_a, q_b = FeatureQueue(slots=4, slot_bytes=2 << 20, ctx=ctx), FeatureQueue(slots=4, slot_bytes=2 << 20, ctx=ctx)
    p = ctx.Process(target=_raw_echo, args=(q_a, q_b, 3), daemon=True)
    p.start()
    rng = numpy.random.default_rng(5)
    arrays = [rng.normal(size=(300, 9)).astype(numpy.float32), rng.random((300, 1)) < 0.5, rng.normal(size=24000), numpy.zeros((0, 513), numpy.float32),
              numpy.arange(7, dtype=numpy.int32)]
    q_a.put_arrays(1, (5, -6, 1 << 40), arrays)
    q_a.put({'kind': 'ready'})
    q_a.put_arrays(9, (), [arrays[0][10:20]])                # a row slice stays contiguous
    m = q_b.get(timeout=120)
    assert isinstance(m, transport.Raw) and m.tag == 2 and m.ints == (10, -12, 1 << 41)
    for got, want in zip(m.arrays, arrays):
        assert got.dtype == want.dtype and got.shape == want.shape and numpy.array_equal(got, want)
    assert q_b.get(timeout=120) == {'kind': 'ready'}
    m = q_b.get(timeout=120)
    assert m.tag == 10 and m.ints == () and numpy.array_equal(m.arrays[0], arrays[0][10:20])
    p.join(timeout=30)
    assert p.exitcode == 0
    with pytest.raises(ValueError, match='slot_bytes'):
        FeatureQueue(slots=1, slot_bytes=4096).put_arrays(0, (), [numpy.zeros(4096, numpy.float32)])
    q_a.close(); q_b.close()
