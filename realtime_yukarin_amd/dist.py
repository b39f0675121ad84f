"""Chunk-level data parallelism over the GPUs of one node.

Each `ConvertStream.process` window is a pure function of its fetched block (overlap context is copied into the
window by `fetch(extra_time)`: /root/reference/realtime_voice_conversion/stream/base_stream.py:38-40, and dropped
afterwards: stream/convert_stream.py:40-42), so windows shard across ranks with NO data-path collective:
window i -> rank i mod world.  The only communication is one broadcast of each predictor's flat weight blob from
rank 0 at start-up (RCCL over xGMI), and an optional gather of the results in window order -- the order run.py
re-establishes with `Item.index` (/root/reference/run.py:171-183).  One process per GPU.

Two interchangeable transports for that broadcast:
* `TorchComm`  -- `torch.distributed` (backend "nccl" = RCCL on ROCm; "gloo" on CPU in the tests): what `bench.py` uses under
  `python -m torch.distributed.run`; torch is plumbing only (process group + one device buffer per blob);
* `NativeComm` -- RCCL through the C ABI (`ry_comm_*` of include/ry355.h, bound with dlopen inside libry355.so): no tensor library
  at all; the 128-byte RCCL id travels from rank 0 to the others through a file next to MASTER_PORT (same node).
"""
import os
import tempfile
import time
from typing import List, Optional, Sequence

import numpy

from . import _lib, engine
from .netspec import NetDesc, param_count
from .weights import flatten_params


# ------------------------------------------------------------------------------------------------ torch.distributed transport
def _dist():
    import torch.distributed as dist
    return dist


def world() -> int:
    try:
        dist = _dist()
    except ImportError:                      # the torch-free transport (NativeComm) must not need torch for its helpers
        return 1
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    try:
        dist = _dist()
    except ImportError:
        return 0
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def broadcast_blob(desc: NetDesc, params: Optional[dict], device):
    """Flat weight blob on `device` on every rank; only rank 0 needs `params`.  One collective per predictor."""
    import torch
    dist = _dist()
    n = param_count(desc)
    if rank() == 0:
        if params is None:
            raise ValueError('rank 0 must provide the weights')
        t = torch.from_numpy(flatten_params(desc, params)).to(device)
    else:
        t = torch.empty(n, dtype=torch.float32, device=device)
    if dist.is_available() and dist.is_initialized():        # also with ONE rank: the collective then really runs (RCCL init, load order)
        dist.broadcast(t, src=0)
    return t


def make_net(ctx: engine.Context, desc: NetDesc, blob, width: int = 512) -> engine.Net:
    """Adopt a broadcast blob: device tensors are handed over by pointer, CPU tensors (gloo tests) as arrays, (ptr, n) as is."""
    if isinstance(blob, tuple):
        return engine.Net(ctx, desc, blob, width=width)
    if blob.is_cuda:
        import torch
        torch.cuda.synchronize(blob.device)
        return engine.Net(ctx, desc, (blob.data_ptr(), blob.numel()), width=width)
    return engine.Net(ctx, desc, blob.numpy(), width=width)


class TorchComm(object):
    """The pieces bench.py needs from a process group, over torch.distributed."""

    def __init__(self, backend: str, rank_: int, world_: int, device=None):
        import torch
        dist = _dist()
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29577')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        kw = {}
        if backend == 'nccl' and device is not None:
            kw['device_id'] = device
        dist.init_process_group(backend, rank=rank_, world_size=world_, **kw)
        self.device = device if device is not None else torch.device('cpu')
        self.rank, self.world, self.kind = rank_, world_, 'torch.distributed/' + backend

    def broadcast_net(self, ctx, desc, params, width=512):
        return make_net(ctx, desc, broadcast_blob(desc, params, self.device), width=width)

    @property
    def ranks_seen(self) -> int:
        return int(_dist().get_world_size())

    def barrier(self):
        _dist().barrier()

    def max(self, v: float) -> float:
        import torch
        dist = _dist()
        t = torch.tensor([v], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        dist = _dist()
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ RCCL through the C ABI
_comm_serial = [0]         # NativeComm objects built so far in this process: every rank builds them in the same order


def _launcher_identity() -> str:
    """`<pid>-<start time in clock ticks since boot>` of the process that launched the ranks (their common parent): unique per launch --
    a recycled pid has another start time -- and the same for every rank, read from /proc/<ppid>/stat field 22.  (Round 3 used the mtime
    of /proc/<ppid> as the launcher's start time; procfs stamps that inode at first lookup, so it was "roughly now", and a rank that
    arrived more than two seconds after rank 0 wrote the id rejected it for the whole timeout.)  RY_COMM_NONCE, when a launcher sets it,
    replaces the whole thing."""
    nonce = os.environ.get('RY_COMM_NONCE')
    if nonce:
        return nonce
    ppid = os.getppid()
    try:
        with open('/proc/%d/stat' % ppid, 'rb') as f:
            fields = f.read().rsplit(b')', 1)[1].split()           # the command name may hold blanks and brackets: count from the last ')'
        return '%d-%s' % (ppid, fields[19].decode())                # field 22 (1-based) = index 19 after (pid, comm)
    except (OSError, IndexError):
        return '%d-0' % ppid


def _proc_start(pid: int) -> Optional[str]:
    """start time of process `pid` in clock ticks since boot (/proc/<pid>/stat field 22), None when there is no such process"""
    try:
        with open('/proc/%d/stat' % pid, 'rb') as f:
            return f.read().rsplit(b')', 1)[1].split()[19].decode()
    except (OSError, IndexError):
        return None


def _id_record(idb: bytes) -> bytes:
    """what rank 0 writes: the 128-byte RCCL id + 32 bytes naming the writer (its pid and start time)"""
    tag = ('%d %s' % (os.getpid(), _proc_start(os.getpid()) or '0')).encode()
    return bytes(idb) + tag.ljust(32, b' ')


def _id_record_is_live(rec: bytes) -> bool:
    """A record counts only while its WRITER is still that very process (same pid, same start time: not a recycled pid): the name of the
    rendezvous file is unique per launcher, but one launcher may start its ranks twice (torchrun --max-restarts, a test process, a
    long-lived shell), and a file that a crashed attempt left behind carries a dead RCCL id -- a rank that picked it up would sit in
    ry_comm_init until the RCCL timeout (round-4 advisor).  The writer of a leftover is dead, so the readers skip it until the new
    rank 0 has replaced it."""
    if len(rec) != 160:
        return False
    try:
        pid_s, start = rec[128:].decode().split()
        if start == '0' or _proc_start(os.getpid()) is None:
            return True              # the check cannot be made (the writer could not read /proc, or this reader cannot): accept, as before the check existed (round-5 advisor)
        seen = _proc_start(int(pid_s))
        if seen is None and not os.path.isdir('/proc/%d' % os.getppid()):
            return True              # another PID namespace than the launcher's (containers sharing the runtime directory): /proc says nothing about the writer
        return seen == start
    except (ValueError, UnicodeDecodeError):
        return False


def _rendezvous_path() -> str:
    """Where rank 0 leaves the RCCL id for the other ranks of the same launch: a file in a directory only this user can write (0700,
    ownership and mode checked), named after the launcher's identity (pid + start time: no other launcher, past or concurrent, has it), the
    elastic restart count when torchrun sets one, MASTER_PORT and the serial number of the communicator within the launch; the record in it
    names its writer, and a reader accepts it only while that writer is alive (_id_record_is_live).  RY_COMM_RENDEZVOUS names the file
    explicitly (the dispatcher passes a path inside a fresh private directory)."""
    p = os.environ.get('RY_COMM_RENDEZVOUS')
    if p:
        return p
    d = os.path.join(tempfile.gettempdir(), 'ry355-%d' % os.getuid())
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.stat(d)
    if st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise _lib.Ry355Error('%s is not a private directory of this user: refusing to exchange the RCCL id through it' % d)
    attempt = '%s.%s' % (os.environ.get('TORCHELASTIC_RUN_ID', ''), os.environ.get('TORCHELASTIC_RESTART_COUNT', '0'))
    attempt = ''.join(c if c.isalnum() or c in '.-' else '_' for c in attempt)[:48]
    return os.path.join(d, 'comm_%s_%s_%s_%d' % (_launcher_identity(), attempt, os.environ.get('MASTER_PORT', '0'), _comm_serial[0]))


class NativeComm(object):
    def __init__(self, ctx: engine.Context, rank_: int, world_: int, timeout: float = 120.0, path: Optional[str] = None):
        import ctypes
        self.ctx, self.rank, self.world, self.kind = ctx, int(rank_), int(world_), 'rccl (C ABI, dlopen)'
        lib = ctx.lib
        idb = ctypes.create_string_buffer(128)
        path = path or _rendezvous_path()
        _comm_serial[0] += 1
        if self.rank == 0:
            lib.check(lib.dll.ry_comm_unique_id(idb))
            if self.world > 1:
                for stale in (path, path + '.tmp'):            # whatever a crashed run (or anybody else) left under this name goes first
                    try:
                        os.remove(stale)
                    except OSError:
                        pass
                fd = os.open(path + '.tmp', os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
                with os.fdopen(fd, 'wb') as f:
                    f.write(_id_record(idb.raw))
                os.replace(path + '.tmp', path)
        else:
            t0 = time.time()
            while True:
                try:
                    st = os.stat(path)
                    if st.st_size == 160 and st.st_uid == os.getuid():    # rank 0 renames the complete record into place ...
                        with open(path, 'rb') as f:
                            rec = f.read(160)
                        if _id_record_is_live(rec):                       # ... and is still the process that wrote it (not the leftover of a crashed attempt)
                            break
                except OSError:
                    pass
                if time.time() - t0 > timeout:
                    raise _lib.Ry355Error('rank %d: no RCCL id of this launch at %s after %.0f s' % (self.rank, path, timeout))
                time.sleep(0.01)
            idb.raw = rec[:128]
        h = ctypes.c_void_p()
        lib.check(lib.dll.ry_comm_init(ctx.handle, idb, self.rank, self.world, ctypes.byref(h)))
        self.handle = h
        self.barrier()
        if self.rank == 0 and self.world > 1:
            try:
                os.remove(path)
            except OSError:
                pass

    def broadcast_net(self, ctx, desc, params, width=512):
        n = param_count(desc)
        ptr = ctx.dev_alloc(n)
        if self.rank == 0:
            if params is None:
                raise ValueError('rank 0 must provide the weights')
            ctx.dev_upload(ptr, flatten_params(desc, params))
        ctx.lib.check(ctx.lib.dll.ry_comm_bcast_weights(self.handle, _lib._fptr(ptr), n, 0))
        net = engine.Net(ctx, desc, (ptr, n), width=width)
        ctx.dev_free(ptr)                                      # ry_net_create re-laid the filters out into its own buffers
        return net

    @property
    def ranks_seen(self) -> int:
        """the rank count RCCL was initialised with and every collective since has completed on (ry_comm_init's barrier included)"""
        return self.world

    def barrier(self):
        self.ctx.lib.check(self.ctx.lib.dll.ry_comm_barrier(self.handle))

    def max(self, v: float) -> float:
        import ctypes
        x = ctypes.c_double(float(v))
        self.ctx.lib.check(self.ctx.lib.dll.ry_comm_allreduce_max(self.handle, ctypes.byref(x)))
        return float(x.value)

    def close(self):
        if self.handle is not None:
            self.barrier()
            self.ctx.lib.dll.ry_comm_destroy(self.handle)
        self.handle = None


# ------------------------------------------------------------------------------------------------ sharding helpers
def shard(n_windows: int, r: Optional[int] = None, w: Optional[int] = None) -> List[int]:
    """Indices of the windows this rank converts (round robin, like a dispatcher handing out Item.index)."""
    r = rank() if r is None else r
    w = world() if w is None else w
    return list(range(r, n_windows, w))


def convert_windows(net: engine.Net, windows: Sequence[numpy.ndarray]) -> List[numpy.ndarray]:
    """Convert this rank's windows; equal-length windows go through the GPU as one batch."""
    if not windows:
        return []
    if len({w.shape for w in windows}) == 1:
        return list(net.convert(numpy.stack(windows)))
    return [net.convert(w) for w in windows]


def gather_in_order(local: List[numpy.ndarray], n_windows: int, dst: int = 0) -> Optional[List[numpy.ndarray]]:
    """Results of all ranks re-assembled in window order on `dst` (None elsewhere)."""
    if world() == 1:
        return list(local)
    dist = _dist()
    bucket = [None] * world() if rank() == dst else None
    dist.gather_object(local, bucket, dst=dst)
    if rank() != dst:
        return None
    out: List[Optional[numpy.ndarray]] = [None] * n_windows
    for r, items in enumerate(bucket):
        for idx, item in zip(shard(n_windows, r, world()), items):
            out[idx] = item
    return out
