"""Worker-side hooks of tests/test_dispatch.py, in a module of their own: a spawned worker process imports the module its hook lives in,
and this one pulls in neither torch nor the oracle (seconds per worker)."""
import time

from realtime_yukarin_amd import engine


def emu_hook(rank):
    """Runs inside every worker process (tests only): the lazily created context becomes the emulator build, and worker 0 is slow."""
    from realtime_yukarin_amd import _lib, build
    ctx = engine.Context(0, _lib.Ry355Lib(build.build_emu()))
    engine.get_context = lambda device=0, lib=None: ctx

    def per_window(index):
        if rank == 0:
            time.sleep(0.2)
    return per_window


def dying_hook(rank):
    emu_hook(rank)
    if rank == 1:
        raise RuntimeError('worker %d cannot see its GPU' % rank)


def failing_window_hook(rank):
    """Worker 0 fails on its second window (after start-up): the dispatcher must report it to the caller instead of waiting for ever."""
    emu_hook(rank)
    seen = []

    def per_window(index):
        seen.append(index)
        if rank == 0 and len(seen) == 2:
            raise RuntimeError('window %d: device lost' % index)
    return per_window


def stall_hook(rank):
    """Worker 0 stalls for two seconds on its second window (a GPU that builds a new launch plan / captures graphs mid-stream): with a
    backlog behind it every ring fills up.  No emulator context: for null workers."""
    seen = []

    def per_window(index):
        seen.append(index)
        if rank == 0 and len(seen) == 2:
            time.sleep(2.0)
    return per_window


def jitter_hook(rank):
    """Eight workers of different, changing speed (no emulator context: for null workers): worker r sleeps 0 .. 35 ms per window by a fixed pseudo-random
    sequence of its own, and every worker has one long stall somewhere in its first windows -- completion order is far from submission order."""
    state = [0x9E3779B9 * (rank + 1) & 0xFFFFFFFF, 0]

    def per_window(index):
        state[0] = (state[0] * 1664525 + 1013904223) & 0xFFFFFFFF
        state[1] += 1
        time.sleep((state[0] >> 24) % 36 / 1000.0 + (0.25 if state[1] == 2 + rank % 3 else 0.0))
    return per_window


def dying_mid_stream_hook(rank):
    """Worker 5 of eight is lost on its third window (null workers)."""
    seen = []

    def per_window(index):
        seen.append(index)
        if rank == 5 and len(seen) == 3:
            raise RuntimeError('window %d: device lost' % index)
    return per_window


def long_stall_hook(rank):
    """Worker 3 sleeps 8 s on its first window (null workers): close() must not wait for it beyond its own deadline."""
    def per_window(index):
        if rank == 3:
            time.sleep(8.0)
    return per_window
