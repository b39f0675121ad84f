"""N > 1 path on CPU: two gloo processes, weights broadcast from rank 0, windows sharded round robin, each rank
converting on the emulator build of the kernels, results gathered in window order and checked against the oracle."""
import os
import socket
import sys
from pathlib import Path

import numpy
import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
N_WINDOWS = 5


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from realtime_yukarin_amd import _lib, build, dist as rdist, engine, synth
    from realtime_yukarin_amd.netspec import NetDesc
    from realtime_yukarin_amd.weights import synthetic_params
    ctx = engine.Context(0, _lib.Ry355Lib(build.EMU_LIB))
    d1, d2 = NetDesc(1, 9, 9, 8, 8), NetDesc(2, 1, 1, 8, 8)
    P1 = synthetic_params(d1, 31) if rank == 0 else None          # only rank 0 owns the weights
    P2 = synthetic_params(d2, 32) if rank == 0 else None
    dev = torch.device('cpu')
    n1 = rdist.make_net(ctx, d1, rdist.broadcast_blob(d1, P1, dev))
    n2 = rdist.make_net(ctx, d2, rdist.broadcast_blob(d2, P2, dev), width=128)
    x = synth.stage1_input(40, N_WINDOWS, seed=77)                 # every rank can see all inputs; it converts only its shard
    sp = synth.stage2_input(40, N_WINDOWS, seed=78, bins=129)
    mine = rdist.shard(N_WINDOWS)
    y1 = rdist.convert_windows(n1, [x[i] for i in mine])
    y2 = rdist.convert_windows(n2, [sp[i] for i in mine])
    g1 = rdist.gather_in_order(y1, N_WINDOWS)
    g2 = rdist.gather_in_order(y2, N_WINDOWS)
    if rank == 0:
        numpy.savez(os.path.join(out_dir, 'gathered.npz'), y1=numpy.stack(g1), y2=numpy.stack(g2), shard0=numpy.array(mine))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_chunk_parallel_convert(tmp_path):
    from realtime_yukarin_amd import build
    build.build_emu()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    z = numpy.load(tmp_path / 'gathered.npz')
    assert list(z['shard0']) == [0, 2, 4]
    from oracle import unet
    from realtime_yukarin_amd import synth
    from realtime_yukarin_amd.netspec import NetDesc
    from realtime_yukarin_amd.weights import synthetic_params
    P1, P2 = synthetic_params(NetDesc(1, 9, 9, 8, 8), 31), synthetic_params(NetDesc(2, 1, 1, 8, 8), 32)
    x = synth.stage1_input(40, N_WINDOWS, seed=77)
    sp = synth.stage2_input(40, N_WINDOWS, seed=78, bins=129)
    for i in range(N_WINDOWS):
        r1 = unet.stage1_convert_core(x[i], P1)
        assert numpy.abs(z['y1'][i] - r1).max() / numpy.abs(r1).max() < 1e-4
        assert numpy.abs(z['y2'][i] / unet.stage2_convert(sp[i], P2) - 1).max() < 1e-4


def test_shard_is_a_partition():
    from realtime_yukarin_amd import dist as rdist
    for w in (1, 2, 4, 8):
        seen = sorted(i for r in range(w) for i in rdist.shard(11, r, w))
        assert seen == list(range(11))
