"""Shared-memory ring transport for the feature items that travel between the encode / convert / decode workers
(SURVEY.md section 8(f) row 3).

What it replaces: the reference builds four `multiprocessing.Queue()`s (/root/reference/run.py:48-51) and sends whole
`Item(item=AcousticFeatureWrapper | AcousticFeature | ndarray, index=int)` objects through them
(realtime_voice_conversion/worker/utility.py:6-13, convert_worker.py:46,57).  `Queue.put` pickles the object in a feeder
thread, writes the bytes through a pipe, and `get` reads and unpickles them: for a 300-frame window (ap + sp = 2 x (300, 513)
floats plus the raw wave) that is three copies of ~1.3-2.5 MB and two thread hand-offs per hop, the same order as the
1.9 ms the whole GPU path takes for that window.

`FeatureQueue` keeps the `multiprocessing.Queue` calling convention (`put`, `get`, `put_nowait`, `get_nowait`, `empty`,
`full`, `qsize`, `close`; `queue.Empty` / `queue.Full`), so the swap in run.py is the constructor call only.  Any
picklable object is accepted; with pickle protocol 5 every C- or F-contiguous ndarray inside it is taken out of band
and written once, 64-byte aligned, into a slot of one shared-memory block, and the small in-band remainder (class
references, scalars, shapes) goes in the slot header.  The receiver copies the slot payload once into private memory
and rebuilds the object around views of that copy, so it owns what it gets (the convert worker keeps input items in its
stream for later overlap windows, convert_stream.py:33-42) and the slot is free again as soon as `get` returns.

Synchronisation: two counting semaphores (free / filled slots) and one lock per side; slots are written and read in
ring order, so delivery is FIFO with any number of producers and consumers.  Like `multiprocessing.Queue`, an
instance must reach a child process as a `Process(...)` argument (fork inherits it, spawn pickles it)."""
import atexit
import os
import pickle
import queue
import struct
import threading
import time
from multiprocessing import get_context, resource_tracker, shared_memory

import collections

import numpy

ALIGN = 64
_ATTACH_LOCK = threading.Lock()
_HEAD = struct.Struct('<QI')      # in-band length, number of out-of-band buffers (top bit: the slot holds a raw array message)
_LEN = struct.Struct('<Q')
_RAW = 1 << 31

# A message of plain arrays that skips pickle altogether (`put_arrays`): what the multi-GPU dispatcher sends per window.  `get` returns it
# as this tuple; the arrays are views of the receiver's private copy of the slot.
Raw = collections.namedtuple('Raw', 'tag ints arrays')


def _round_up(n: int) -> int:
    return (n + ALIGN - 1) // ALIGN * ALIGN


def _attach(name: str) -> shared_memory.SharedMemory:
    """Attach to the creator's block WITHOUT registering it with the resource tracker.  Before Python 3.13 an attach registers the
    name as if this process had created it, so the tracker would unlink the block when the attaching process exits.  Unregistering
    afterwards is no cure: fork / spawn children share the creator's tracker process, whose registry is a set, so the child's
    unregister also drops the CREATOR's entry and the creator's own `unlink()` then trips a KeyError inside the tracker.  Only the
    creating process owns the name; attaching leaves the registry alone."""
    try:
        return shared_memory.SharedMemory(name=name, track=False)            # Python >= 3.13
    except TypeError:
        pass
    with _ATTACH_LOCK:                                    # the patch is process-global: one attaching thread at a time
        real = resource_tracker.register
        resource_tracker.register = lambda *a, **k: None
        try:
            return shared_memory.SharedMemory(name=name)
        finally:
            resource_tracker.register = real


class FeatureQueue(object):
    def __init__(self, slots: int = 8, slot_bytes: int = 16 << 20, ctx=None) -> None:
        if slots < 1 or slot_bytes < 4096:
            raise ValueError('FeatureQueue needs slots >= 1 and slot_bytes >= 4096')
        ctx = ctx or get_context()
        self.slots = int(slots)
        self.slot_bytes = _round_up(int(slot_bytes))
        self._shm = shared_memory.SharedMemory(create=True, size=self.slots * self.slot_bytes)
        self._owner = os.getpid()
        self._free = ctx.Semaphore(self.slots)
        self._filled = ctx.Semaphore(0)
        self._put_lock = ctx.Lock()
        self._get_lock = ctx.Lock()
        self._head = ctx.RawValue('Q', 0)          # items ever written   (guarded by _put_lock)
        self._tail = ctx.RawValue('Q', 0)          # items ever read      (guarded by _get_lock)
        atexit.register(self.close)

    # ---- reaching a child process -----------------------------------------------------------------------------------
    def __getstate__(self):
        return dict(slots=self.slots, slot_bytes=self.slot_bytes, name=self._shm.name, owner=self._owner, free=self._free,
                    filled=self._filled, put_lock=self._put_lock, get_lock=self._get_lock, head=self._head, tail=self._tail)

    def __setstate__(self, s):
        self.slots, self.slot_bytes, self._owner = s['slots'], s['slot_bytes'], s['owner']
        self._free, self._filled, self._put_lock, self._get_lock = s['free'], s['filled'], s['put_lock'], s['get_lock']
        self._head, self._tail = s['head'], s['tail']
        self._shm = _attach(s['name'])

    # ---- producer ---------------------------------------------------------------------------------------------------
    def put(self, obj, block: bool = True, timeout=None) -> None:
        bufs = []
        inband = pickle.dumps(obj, protocol=5, buffer_callback=bufs.append)
        raws = [b.raw() for b in bufs]
        head_len = _round_up(_HEAD.size + _LEN.size * len(raws) + len(inband))
        total = head_len + sum(_round_up(r.nbytes) for r in raws)
        if total > self.slot_bytes:
            raise ValueError('item needs %d bytes, FeatureQueue slots hold %d (raise slot_bytes)' % (total, self.slot_bytes))
        if not self._free.acquire(block, timeout):
            raise queue.Full
        with self._put_lock:
            base = (self._head.value % self.slots) * self.slot_bytes
            mem = self._shm.buf
            _HEAD.pack_into(mem, base, len(inband), len(raws))
            off = base + _HEAD.size
            for r in raws:
                _LEN.pack_into(mem, off, r.nbytes)
                off += _LEN.size
            mem[off:off + len(inband)] = inband
            off = base + head_len
            for r in raws:
                mem[off:off + r.nbytes] = r
                off += _round_up(r.nbytes)
            self._head.value += 1
        self._filled.release()

    def put_nowait(self, obj) -> None:
        self.put(obj, False)

    def put_arrays(self, tag: int, ints, arrays, block: bool = True, timeout=None) -> None:
        """A message of one tag, a few integers and C-contiguous ndarrays, written WITHOUT pickle: a fixed little descriptor (dtype, shape) per
        array in the slot header, the array bytes 64-byte aligned behind it.  `get` hands it back as `Raw(tag, ints, arrays)`.  For the
        hot messages of the dispatcher: pickling the feature object and its five arrays out of band costs ~30 + 20 us per hop, this ~8."""
        arrays = [numpy.ascontiguousarray(a) for a in arrays]
        desc = [struct.pack('<qI', int(tag), len(ints)), struct.pack('<%dq' % len(ints), *[int(v) for v in ints])]
        for a in arrays:
            desc.append(struct.pack('<4sB%dI' % a.ndim, a.dtype.str.encode().ljust(4), a.ndim, *a.shape))
        inband = b''.join(desc)
        head_len = _round_up(_HEAD.size + _LEN.size * len(arrays) + len(inband))
        total = head_len + sum(_round_up(a.nbytes) for a in arrays)
        if total > self.slot_bytes:
            raise ValueError('message needs %d bytes, FeatureQueue slots hold %d (raise slot_bytes)' % (total, self.slot_bytes))
        if not self._free.acquire(block, timeout):
            raise queue.Full
        with self._put_lock:
            base = (self._head.value % self.slots) * self.slot_bytes
            mem = self._shm.buf
            _HEAD.pack_into(mem, base, len(inband), len(arrays) | _RAW)
            off = base + _HEAD.size
            for a in arrays:
                _LEN.pack_into(mem, off, a.nbytes)
                off += _LEN.size
            mem[off:off + len(inband)] = inband
            off = base + head_len
            for a in arrays:
                if a.nbytes:
                    mem[off:off + a.nbytes] = a.reshape(-1).view(numpy.uint8).data
                off += _round_up(a.nbytes)
            self._head.value += 1
        self._filled.release()

    # ---- consumer ---------------------------------------------------------------------------------------------------
    def get(self, block: bool = True, timeout=None):
        if not self._filled.acquire(block, timeout):
            raise queue.Empty
        with self._get_lock:
            base = (self._tail.value % self.slots) * self.slot_bytes
            mem = self._shm.buf
            n_inband, n_bufs = _HEAD.unpack_from(mem, base)
            raw, n_bufs = bool(n_bufs & _RAW), n_bufs & ~_RAW
            off = base + _HEAD.size
            lens = [_LEN.unpack_from(mem, off + i * _LEN.size)[0] for i in range(n_bufs)]
            off += _LEN.size * n_bufs
            inband = bytes(mem[off:off + n_inband])
            head_len = _round_up(_HEAD.size + _LEN.size * n_bufs + n_inband)
            payload = bytearray(mem[base + head_len:base + head_len + sum(_round_up(n) for n in lens)])   # the one copy
            self._tail.value += 1
        self._free.release()
        view, views, off = memoryview(payload), [], 0
        for n in lens:
            views.append(view[off:off + n])
            off += _round_up(n)
        if not raw:
            return pickle.loads(inband, buffers=views)
        tag, n_ints = struct.unpack_from('<qI', inband, 0)
        ints = struct.unpack_from('<%dq' % n_ints, inband, 12)
        pos, arrays = 12 + 8 * n_ints, []
        for v in views:
            dt, ndim = struct.unpack_from('<4sB', inband, pos)
            shape = struct.unpack_from('<%dI' % ndim, inband, pos + 5)
            pos += 5 + 4 * ndim
            arrays.append(numpy.frombuffer(v, dtype=numpy.dtype(dt.decode().strip())).reshape(shape))
        return Raw(tag, ints, arrays)

    def get_nowait(self):
        return self.get(False)

    # ---- state ------------------------------------------------------------------------------------------------------
    def qsize(self) -> int:
        return int(self._head.value - self._tail.value)

    def empty(self) -> bool:
        return self.qsize() <= 0

    def full(self) -> bool:
        return self.qsize() >= self.slots

    def close(self) -> None:
        """Detach; the creating process also removes the block (idempotent, registered with atexit there)."""
        shm, self._shm = getattr(self, '_shm', None), None
        if shm is None:
            return
        try:
            shm.close()
        except BufferError:          # a caller still holds a view of the block; the OS frees it with the process
            pass
        if os.getpid() == self._owner:
            try:
                shm.unlink()
            except FileNotFoundError:
                pass

    def join_thread(self) -> None:   # multiprocessing.Queue API: there is no feeder thread here
        pass

    def cancel_join_thread(self) -> None:
        pass


def echo_worker(q_in, q_out, n: int) -> None:
    """Child side of the round-trip measurement (`measure_round_trip`): get an item, send it back."""
    for _ in range(n):
        q_out.put(q_in.get())


def measure_round_trip(make_queue, obj, n: int = 50, warmup: int = 5, ctx=None) -> float:
    """Median seconds for parent -> child -> parent of `obj` over two queues built by `make_queue()`; the same harness
    times `multiprocessing.Queue` and `FeatureQueue` (scripts/transport_bench.py)."""
    ctx = ctx or get_context()
    q_a, q_b = make_queue(), make_queue()
    p = ctx.Process(target=echo_worker, args=(q_a, q_b, n + warmup), daemon=True)
    p.start()
    times = []
    for i in range(n + warmup):
        t0 = time.perf_counter()
        q_a.put(obj)
        q_b.get(timeout=60)
        if i >= warmup:
            times.append(time.perf_counter() - t0)
    p.join(timeout=10)
    for q in (q_a, q_b):
        q.close()
    times.sort()
    return times[len(times) // 2]
