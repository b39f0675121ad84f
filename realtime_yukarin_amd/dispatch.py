"""Multi-GPU dispatcher of the CHAINED convert path: one stream, G GPUs, window i -> GPU i mod G, results released in `Item.index` order.

north_star: "independent buffer_time chunks are sharded across the 8 GPUs of one node with RCCL broadcast of weights over xGMI and no
cross-chunk collectives" (BASELINE.json).  What makes that legal is the reference's own structure: `ConvertStream.process` copies the
overlap context INTO the window it fetches (/root/reference/realtime_voice_conversion/stream/base_stream.py:32-79: `fetch(start_time,
time_length, extra_time)`) and drops it afterwards (stream/convert_stream.py:40-42), so a window is a pure function of what `fetch`
returned; the consumer re-establishes the order by `Item.index` (/root/reference/run.py:171-183).  So:

    queue_input --> [dispatcher: ONE ConvertStream: add / fetch / remove] --window k--> worker k mod G (one process per GPU:
                                                                                        its own HIP context, VoiceChanger, window core)
    queue_output <-- [dispatcher: release in index order] <----- (index, picked feature) ---- any worker, any order

* the windows travel through `transport.FeatureQueue` rings (shared memory, one copy per hop; one ring into every worker and one ring
  back from every worker) -- no pickled `gather_object`, no second copy through a pipe; a window travels WITHOUT its `ap` block, which
  the stage passes through untouched: it stays in the dispatcher and is re-attached on release;
* the weights travel ONCE: worker 0 unpickles the converter objects with their host weights (exactly what the reference ships to its
  single convert worker, /root/reference/run.py:69-79), workers 1 .. G-1 get copies WITHOUT weights (`without_weights`) and receive
  both predictors by one RCCL broadcast each over xGMI (`dist.NativeComm`: `ry_comm_bcast_weights`, include/ry355.h), then `adopt_net`;
  `comm='host'` ships full copies instead (no RCCL: the CPU tests on the emulator, or a single GPU);
* no collective after start-up; every worker keeps `depth` windows in flight on its pinned ring (`VoiceChanger.begin` / `finish`) and
  announces the frames the stream's `pick` throws away (`discard`), so stage 2 does not compute them;
* results are released strictly in submission order (= `Item.index` order of run.py), whatever order the GPUs finish in.

`convert_worker_multi_gpu` is the drop-in for the reference's `convert_worker` process target (same arguments + `devices`).
In the LIVE path windows arrive one per buffer_time, so G GPUs raise throughput (offline conversion, a backlog, many sessions), not the
latency of one window."""
import collections
import logging
import os
import queue
import shutil
import tempfile
import time
import traceback
from multiprocessing import get_context
from typing import Callable, List, Optional, Sequence, Tuple

import numpy

from . import transport

_STOP = None
_TAG_WIN, _TAG_OK = 1, 2                             # raw array messages (transport.put_arrays): a lean window in, its lean result back
PICK_KEYS = ('f0', 'ap', 'sp', 'voiced')             # FeatureSegmentMethod._keys (segment/feature_segment.py:21): what ConvertStream.process returns
FULL_KEYS = ('f0', 'ap', 'voiced', 'mc', 'sp')       # what convert_from_acoustic_feature sets (combine_silent's keys + sp)


def _rows(n: int, pick) -> Tuple[int, int]:
    """[first, last) of the frames `feature.pick(pick[0], pick[1], ...)` keeps of an n-frame window (python slice rules)."""
    if pick is None:
        return 0, n
    k0, k1, _ = slice(pick[0], pick[1]).indices(n)
    return k0, max(k0, k1)


def _pace_hook(ms: float, rank: int):
    """Worker hook of the paced null workers (bench.py --dispatcher): every window takes `ms` of busy waiting, a stand-in for the GPU time."""
    def per_window(seq):
        t_end = time.perf_counter() + ms * 1e-3
        while time.perf_counter() < t_end:
            pass
    return per_window


def paced(ms: float) -> Callable:
    import functools
    return functools.partial(_pace_hook, float(ms))


def _worker_main(rank: int, world: int, device: int, ac, sr, threshold, comm: str, rendezvous: str, depth: int,
                 q_in, q_out, avail, hook: Optional[Callable], null: bool) -> None:
    """One GPU: build (or receive) the predictors, then convert windows until the stop item.  Every message on q_out is
    (kind, rank, seq, payload): ('ready', r, -1, None), ('ok', r, seq, payload), ('error', r, seq, text); `avail` is released once per
    message (the parent sleeps on it and then looks into the workers' rings).
    payload: lean windows -> (k0, sp[k0:k1], f0[k0:k1], voiced[k0:k1], mc[k0:k1], effective) -- the parent re-attaches `ap`; others ->
    the (picked) feature object."""
    def send(msg):
        q_out.put(msg)
        avail.release()

    def send_lean(seq, k0, sp, f0, voiced, mc, effective):             # the hot message: plain arrays, no pickle (transport.put_arrays)
        q_out.put_arrays(_TAG_OK, (rank, seq, k0), (sp, f0, voiced, mc, effective))
        avail.release()

    def unpack(msg):
        """(seq, window, discard, pick, lean) from either form of a window message."""
        if not isinstance(msg, transport.Raw):
            return msg
        seq, d0, d1, has_pick, p0, p1, rate = msg.ints
        wave, f0, mc, voiced = msg.arrays
        f_in = AcousticFeature(f0=f0, mc=mc, voiced=voiced)           # `ap` stayed with the dispatcher
        f_in.wave = Wave(wave=wave, sampling_rate=int(rate))
        return seq, f_in, (int(d0), int(d1)), ((int(p0), int(p1), ()) if has_pick else None), True
    try:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC for RCCL (before the HIP runtime comes up)
        os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')               # one hardware queue per stream of the window lanes
        from . import compat
        compat.install()
        from yukarin.acoustic_feature import AcousticFeature
        from yukarin.wave import Wave
        if null:                                                        # measurement aid (bench.py --dispatcher): no GPU, no predictors --
            bins = 513                                                  # the worker answers with a payload of the real size, at once or
            per_window = hook(rank) if hook is not None else None      # after the hook's pause (a stand-in for the GPU time of a window)
            send(('ready', rank, -1, None))
            zero = {}
            while True:
                msg = q_in.get()
                if msg is _STOP:
                    break
                seq, f_in, discard, pick, lean = unpack(msg)
                if per_window is not None:
                    per_window(seq)
                n = len(f_in.f0)
                k0, k1 = _rows(n, pick)
                if (n, k0, k1) not in zero:
                    zero[(n, k0, k1)] = (k0, numpy.zeros((k1 - k0, bins), numpy.float32), numpy.zeros((k1 - k0, 1), numpy.float32),
                                         numpy.zeros((k1 - k0, 1), bool), numpy.zeros((k1 - k0, 9), numpy.float32), numpy.ones(n, bool))
                if lean:
                    send_lean(seq, *zero[(n, k0, k1)])
                else:
                    send(('ok', rank, seq, f_in if pick is None else f_in.pick(pick[0], pick[1], keys=list(pick[2]))))
            return
        from . import engine
        per_window = hook(rank) if hook is not None else None         # tests: route the context to the emulator, inject jitter
        ac.gpu = sr.gpu = device
        if comm == 'native':
            from . import dist as rdist
            ctx = engine.get_context(device)
            c = rdist.NativeComm(ctx, rank, world, path=rendezvous)
            n1 = c.broadcast_net(ctx, ac.desc, ac._params if rank == 0 else None)
            bins = ac.mc2sp_matrix().shape[1]
            n2 = c.broadcast_net(ctx, sr.desc, sr._params if rank == 0 else None, width=bins - 1)
            ac.adopt_net(n1); sr.adopt_net(n2)
            c.close()                                                   # no collective after start-up
        from .voice_changer import VoiceChanger
        vc = VoiceChanger(acoustic_converter=ac, super_resolution=sr, threshold=threshold)
        vc._fused_core()                                                # contexts, predictors and the window core exist before 'ready'
        import gc
        gc.collect(); gc.freeze()                                       # a full collection mid-stream is a pause of tens of ms (profiles/r04/driver_cmd.txt)
        send(('ready', rank, -1, None))
        pending = collections.deque()

        def finish_one():
            seq, handle, pick, lean = pending.popleft()
            out = vc.finish(handle, lean=lean)
            if lean and getattr(out, 'effective', None) is not None:
                k0, k1 = _rows(len(out.effective), pick)
                send_lean(seq, k0, out.sp[k0:k1], out.f0[k0:k1], out.voiced[k0:k1], out.mc[k0:k1], out.effective)
                return
            if pick is not None:
                out = out.pick(pick[0], pick[1], keys=list(pick[2]))
            send(('ok', rank, seq, out))
        while True:
            if pending and (len(pending) >= depth or q_in.empty()):
                finish_one()
                continue
            msg = q_in.get()
            if msg is _STOP:
                while pending:
                    finish_one()
                break
            seq, f_in, discard, pick, lean = unpack(msg)
            if per_window is not None:
                per_window(seq)
            pending.append((seq, vc.begin(f_in, discard=discard), pick, lean))
        vc.close(); ac.close(); sr.close()
    except BaseException:                                               # the parent must hear about it: a silent worker death hangs the stream
        try:
            send(('error', rank, -1, traceback.format_exc()))
        except Exception:
            pass
        raise


class ChunkDispatcher(object):
    """G worker processes, one per entry of `devices`; `submit` hands window k to worker k mod G, `collect` returns finished windows in
    submission order.  `comm`: 'native' = weights by RCCL broadcast from worker 0 (`ry_comm_*`), 'host' = every worker unpickles its own
    copy, 'auto' = 'native' when there is more than one device, else 'host'.

    What travels per window (canonical mc -> mc converters): up, the window WITHOUT its `ap` block (the stage passes ap through untouched:
    voice_changer.py:33-37 -- it stays with the dispatcher and is re-attached on release, zeroed on the frames the silence gate cut); down,
    the kept rows of sp / f0 / voiced / mc and the gate's mask.  At 300 frames with 100 kept: 0.11 MB up and 0.21 MB down instead of
    0.73 MB + 0.41 MB.  One ring into every worker and one ring back from every worker (single producer, single consumer each), plus
    one counting semaphore the dispatcher sleeps on."""

    def __init__(self, acoustic_converter, super_resolution, devices: Sequence[int], threshold: float = 60, comm: str = 'auto',
                 depth: int = 2, mp_context: str = 'spawn', worker_hook: Optional[Callable] = None, slots: int = 8,
                 slot_bytes: int = 16 << 20, start_timeout: float = 600.0, null_workers: bool = False, lean: Optional[bool] = None) -> None:
        if not devices:
            raise ValueError('ChunkDispatcher needs at least one device')
        if comm not in ('auto', 'native', 'host'):
            raise ValueError("comm must be 'auto', 'native' or 'host'")
        self.devices = [int(d) for d in devices]
        self.world = len(self.devices)
        self.comm = ('native' if self.world > 1 else 'host') if comm == 'auto' else comm
        # lean windows need the device-resident path in the workers (canonical mc -> mc converters); anything else ships whole objects
        can_lean = bool(getattr(acoustic_converter, 'fusable', lambda: False)()) and hasattr(super_resolution, '_get_net')
        self.lean = can_lean if lean is None else (bool(lean) and (can_lean or null_workers))
        self._bins = None
        self._ac_sizes = getattr(acoustic_converter, '_sizes', None) or (lambda: {})
        self._mp = get_context(mp_context)
        self._q_in = [transport.FeatureQueue(slots, slot_bytes, ctx=self._mp) for _ in self.devices]
        self._q_out = [transport.FeatureQueue(slots, slot_bytes, ctx=self._mp) for _ in self.devices]
        self._avail = self._mp.Semaphore(0)                            # one release per message on any return ring
        self._scan = 0
        self._dir = tempfile.mkdtemp(prefix='ry355-dispatch-')        # 0700: the RCCL id of this dispatcher's workers lives here
        self._procs = []
        self._submitted = 0            # windows handed out so far: window k goes to worker k mod G
        self._released = 0             # windows returned to the caller so far
        self._labels = {}              # sequence number -> (the caller's index, the window's ap, pick)
        self._done = {}                # sequence number -> result, waiting for its turn
        self.max_out_of_order = 0      # how far ahead of the release point a result has arrived (diagnostics / tests)
        self.closed = False
        null = bool(null_workers)
        lean_ac = acoustic_converter.without_weights() if (null or (self.comm == 'native' and self.world > 1)) else None
        lean_sr = super_resolution.without_weights() if (null or (self.comm == 'native' and self.world > 1)) else None
        for r, dev in enumerate(self.devices):
            ac = acoustic_converter if ((r == 0 and not null) or lean_ac is None) else lean_ac
            sr = super_resolution if ((r == 0 and not null) or lean_sr is None) else lean_sr
            p = self._mp.Process(target=_worker_main, name='ry355-gpu%d' % dev, daemon=True,
                                 args=(r, self.world, dev, ac, sr, threshold, self.comm, os.path.join(self._dir, 'rccl_id'), int(depth),
                                       self._q_in[r], self._q_out[r], self._avail, worker_hook, null))
            p.start()
            self._procs.append(p)
        ready, t0 = 0, time.time()
        while ready < self.world:
            kind, r, _, payload = self._get(start_timeout - (time.time() - t0))
            if kind != 'ready':
                self.close()
                raise RuntimeError('worker %d failed to start:\n%s' % (r, payload))
            ready += 1

    # ---- plumbing
    def _next_message(self):
        """The message `_avail` was just acquired for: it is in one of the return rings (the worker puts, then releases)."""
        for _ in range(2 * self.world + 2):
            for k in range(self.world):
                r = (self._scan + k) % self.world
                try:
                    msg = self._q_out[r].get_nowait()
                except queue.Empty:
                    continue
                self._scan = r + 1
                return msg
            time.sleep(0.0005)                                          # (cannot happen: put precedes release)
        raise RuntimeError('dispatcher: a result was announced but none of the return rings holds it')

    def _get(self, timeout: Optional[float]):
        """One message from the workers; a worker that died without a word is an error, not a hang."""
        deadline = None if timeout is None else time.time() + max(timeout, 0.0)
        while True:
            if self._avail.acquire(True, 0.2):
                return self._next_message()
            for r, p in enumerate(self._procs):
                if not p.is_alive() and p.exitcode not in (0, None):
                    return ('error', r, -1, 'worker process %s exited with code %s' % (p.name, p.exitcode))
            if deadline is not None and time.time() > deadline:
                raise TimeoutError('no message from the GPU workers')

    def _take(self, msg) -> None:
        if isinstance(msg, transport.Raw):                              # a lean result: (rank, seq, k0), [sp, f0, voiced, mc, effective]
            r, seq, k0 = msg.ints
            kind, payload = 'ok', (int(k0),) + tuple(msg.arrays)
        else:
            kind, r, seq, payload = msg
        if kind == 'error':
            self.close()
            raise RuntimeError('GPU worker %d failed:\n%s' % (r, payload))
        self._done[seq] = payload
        self.max_out_of_order = max(self.max_out_of_order, seq - self._released)

    def _poll(self) -> None:
        """Take whatever the workers have finished, without waiting."""
        while self._avail.acquire(False):
            self._take(self._next_message())

    def _assemble(self, seq):
        """The caller's feature object from a worker's lean payload + the `ap` block that stayed here."""
        index, ap_in, pick = self._labels.pop(seq)
        payload = self._done.pop(seq)
        if not (isinstance(payload, tuple) and len(payload) == 6):
            return index, payload
        from .voice_changer import VoiceChanger
        from yukarin.acoustic_feature import AcousticFeature
        k0, sp, f0, voiced, mc, effective = payload
        if self._bins is None:                                          # width of a missing `ap` block: the INPUT rate's size (combine_silent's), not the spectrogram's
            try:
                self._bins = int(self._ac_sizes()['ap'])
            except Exception:
                self._bins = sp.shape[1]
        parts = dict(f0=f0, sp=sp, voiced=voiced, mc=mc)
        keys = FULL_KEYS if pick is None else tuple(pick[2])
        if 'ap' in keys:
            parts['ap'] = VoiceChanger.attach_ap(ap_in, effective, k0, k0 + len(sp), self._bins)
        return index, AcousticFeature(**{k: parts[k] for k in keys if k in parts})

    # ---- the caller's side
    def submit(self, index, f_in, discard: Tuple[int, int] = (0, 0), pick: Optional[Tuple[int, int, Sequence[str]]] = None) -> int:
        """Hand a fetched window to the next GPU (round robin).  `discard` = (front, back) frames the caller throws away (not computed by
        stage 2), `pick` = (first, last, keys): the caller gets `feature.pick(first, last, keys)` (only the kept frames travel back).
        While that worker's ring is full the call keeps TAKING finished windows off the return rings (a stalled GPU with a backlog behind
        it must not stop the others from delivering: every ring is bounded), then hands the window over."""
        if self.closed:
            raise RuntimeError('dispatcher is closed')
        seq = self._submitted
        r = seq % self.world
        lean = self.lean and all(isinstance(getattr(f_in, k, None), numpy.ndarray) for k in ('mc', 'f0', 'voiced'))
        if lean:
            self._labels[seq] = (index, f_in.ap, pick)                 # `ap` stays here; re-attached in _assemble
            w = f_in.wave
            ints = (seq, int(discard[0]), int(discard[1]), 0 if pick is None else 1, 0 if pick is None else int(pick[0]),
                    0 if pick is None else int(pick[1]), int(w.sampling_rate))
            arrays = (numpy.asarray(w.wave), f_in.f0, f_in.mc, f_in.voiced)
        else:
            self._labels[seq] = (index, None, pick)
            msg = (seq, f_in, (int(discard[0]), int(discard[1])), pick, False)
        while True:
            try:
                if lean:
                    self._q_in[r].put_arrays(_TAG_WIN, ints, arrays, True, 0.05)
                else:
                    self._q_in[r].put(msg, True, 0.05)
                break
            except queue.Full:
                self._poll()
                if not self._procs[r].is_alive():
                    self.close()
                    raise RuntimeError('GPU worker %d is gone (exit code %s) with its input ring full' % (r, self._procs[r].exitcode))
        self._submitted += 1
        return r

    def pending(self) -> int:
        return self._submitted - self._released

    def collect(self, block: bool = False, timeout: Optional[float] = None) -> List[Tuple[object, object]]:
        """[(index, feature), ...] of every window whose turn has come, in submission order.  block=True waits until at least the next
        window in order is there (if any is pending)."""
        out = []
        while True:
            self._poll()
            while self._released in self._done:
                out.append(self._assemble(self._released))
                self._released += 1
            if out or not block or self.pending() == 0:
                return out
            self._take(self._get(timeout))

    def drain(self, timeout: Optional[float] = None) -> List[Tuple[object, object]]:
        out = []
        while self.pending():
            out += self.collect(block=True, timeout=timeout)
        return out

    def close(self) -> None:
        if self.closed:
            return
        self.closed = True
        # One deadline for all workers, and the return rings keep being emptied while we wait: a worker with a backlog drains it into its
        # bounded return ring before it sees the stop item, and would otherwise block in `put` for ever once that ring is full (round-4
        # advisor: G x 20 s on the error path).
        deadline = time.time() + 20.0
        unsent = [(q, p) for q, p in zip(self._q_in, self._procs) if p.is_alive()]

        def drain():
            while self._avail.acquire(False):
                try:
                    self._next_message()                     # thrown away: the stream is over
                except Exception:
                    return
        while time.time() < deadline and (unsent or any(p.is_alive() for p in self._procs)):
            drain()
            for q, p in list(unsent):
                if not p.is_alive():
                    unsent.remove((q, p)); continue
                try:
                    q.put(_STOP, True, 0.02)
                    unsent.remove((q, p))
                except Exception:
                    pass
            for p in self._procs:
                p.join(timeout=0.02)
        for p in self._procs:
            if p.is_alive():
                p.terminate()                                # this exact child (never by pattern)
                p.join(timeout=2)
        for q in self._q_in + self._q_out:
            q.close()
        shutil.rmtree(self._dir, ignore_errors=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def convert_worker_multi_gpu(acoustic_converter, super_resolution, time_length: float, extra_time: float, input_silent_threshold: float,
                             queue_input, queue_output, acquired_lock, devices: Sequence[int] = (0,), comm: str = 'auto', depth: int = 2,
                             worker_hook: Optional[Callable] = None, mp_context: str = 'spawn', **dispatcher_kwargs) -> None:
    """Drop-in process target for the reference's `convert_worker` (/root/reference/realtime_voice_conversion/worker/convert_worker.py:17-59):
    same arguments and `Item` protocol, the reference's own `ConvertStream` for `add` / `fetch` (imported from the maintainer's installed
    `realtime_voice_conversion` package at call time), the windows converted on `devices`.  An item of `None` ends the loop."""
    from realtime_voice_conversion.stream import ConvertStream
    from .voice_changer import VoiceChanger
    from .worker import retire_time
    logger = logging.getLogger('convert')
    # the central stream only fetches: its VoiceChanger is never asked to convert (no GPU context in this process)
    stream = ConvertStream(voice_changer=VoiceChanger(super_resolution=super_resolution, acoustic_converter=acoustic_converter,
                                                      threshold=input_silent_threshold))
    pad = round(extra_time * stream.in_segment_method.sampling_rate)           # convert_stream.py:40-42
    pick = (pad, -pad, PICK_KEYS) if pad > 0 else None
    disp = ChunkDispatcher(acoustic_converter, super_resolution, devices, threshold=input_silent_threshold, comm=comm, depth=depth,
                           worker_hook=worker_hook, mp_context=mp_context, **dispatcher_kwargs)
    items = {}
    try:
        acquired_lock.release()
        start_time, current = extra_time, 0.0

        def release(block):
            for index, out_feature in disp.collect(block=block):
                item, t0 = items.pop(index)
                item.item = out_feature
                queue_output.put(item)
                logger.debug('%s: %s', item.index, time.time() - t0)
        while True:
            if disp.pending() and queue_input.empty():
                release(block=True)
                continue
            item = queue_input.get()
            if item is None:
                while disp.pending():
                    release(block=True)
                return
            items[item.index] = (item, time.time())
            stream.add(start_time=start_time, data=item.item)
            start_time += time_length
            in_feature = stream.fetch(start_time=current, time_length=time_length, extra_time=extra_time)
            current += time_length
            disp.submit(item.index, in_feature, discard=(pad, pad), pick=pick)
            stream.remove(end_time=retire_time(current, time_length, extra_time))
            release(block=False)
    finally:
        disp.close()
