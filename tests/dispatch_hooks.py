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
