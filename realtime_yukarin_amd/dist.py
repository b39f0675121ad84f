"""Chunk-level data parallelism over the GPUs of one node.

Each `ConvertStream.process` window is a pure function of its fetched block (overlap context is copied into the
window by `fetch(extra_time)`: /root/reference/realtime_voice_conversion/stream/base_stream.py:38-40, and dropped
afterwards: stream/convert_stream.py:40-42), so windows shard across ranks with NO data-path collective:
window i -> rank i mod world.  The only communication is one broadcast of each predictor's flat weight blob from
rank 0 at start-up (RCCL over xGMI with backend "nccl"; gloo on CPU in the tests), and an optional gather of the
results in window order -- the order run.py re-establishes with `Item.index` (/root/reference/run.py:171-183).
One process per GPU; `torch.distributed` is plumbing only.
"""
from typing import List, Optional, Sequence

import numpy
import torch
import torch.distributed as dist

from . import engine
from .netspec import NetDesc, param_count
from .weights import flatten_params


def world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def broadcast_blob(desc: NetDesc, params: Optional[dict], device: torch.device) -> torch.Tensor:
    """Flat weight blob on `device` on every rank; only rank 0 needs `params`.  One collective per predictor."""
    n = param_count(desc)
    if rank() == 0:
        if params is None:
            raise ValueError('rank 0 must provide the weights')
        t = torch.from_numpy(flatten_params(desc, params)).to(device)
    else:
        t = torch.empty(n, dtype=torch.float32, device=device)
    if world() > 1:
        dist.broadcast(t, src=0)
    return t


def make_net(ctx: engine.Context, desc: NetDesc, blob: torch.Tensor, width: int = 512) -> engine.Net:
    """Adopt a broadcast blob: device tensors are handed over by pointer, CPU tensors (gloo tests) as arrays."""
    if blob.is_cuda:
        torch.cuda.synchronize(blob.device)
        return engine.Net(ctx, desc, (blob.data_ptr(), blob.numel()), width=width)
    return engine.Net(ctx, desc, blob.numpy(), width=width)


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
    bucket = [None] * world() if rank() == dst else None
    dist.gather_object(local, bucket, dst=dst)
    if rank() != dst:
        return None
    out: List[Optional[numpy.ndarray]] = [None] * n_windows
    for r, items in enumerate(bucket):
        for idx, item in zip(shard(n_windows, r, world()), items):
            out[idx] = item
    return out
