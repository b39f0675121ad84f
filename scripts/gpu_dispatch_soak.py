#!/usr/bin/env python3
"""Soak of the multi-GPU dispatcher on the real hardware (GPU box): a few thousand windows of CHANGING length (100 .. 600 frames) and silence
pattern through `dispatch.ChunkDispatcher` with two worker processes on the one GPU -- new launch plans and graph captures mid-stream on
either worker, rings that fill up behind them, lean array messages of every size.  Checked all the way: strictly in-order release, every
result finite with zero `ap` / `mc` rows exactly where the returned mask says so, and a fixed probe window that comes back with the same
bits every time it is sent (whichever worker converts it).  Usage: python scripts/gpu_dispatch_soak.py [windows] [workers]"""
import hashlib
import sys
import tempfile
import time
from pathlib import Path

import numpy

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    from realtime_yukarin_amd import dispatch, synth
    n_windows = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    import os
    emu = bool(os.environ.get('RY_SOAK_EMU'))                          # flow check of this script on the CPU emulator (SYN-8, short windows)
    hook = None
    if emu:
        sys.path.insert(0, str(ROOT / 'tests'))
        from dispatch_hooks import emu_hook as hook
    tmp = Path(tempfile.mkdtemp(prefix='ry355-soak-'))
    synth.write_model_files(tmp, 'SYN-8' if emu else 'SYN-64')
    ac, sr = synth.build_converters(tmp)
    rng = numpy.random.default_rng(11)
    lengths = [40, 300, 64] if emu else [100, 300, 300, 300, 200, 400, 600, 137, 300, 257]
    pool = [synth.feature_window(n, 1000 + i, silent_stretch=(i % 2 == 1)) for i, n in enumerate(lengths)]
    probe = synth.feature_window(300, 4242, silent_stretch=True)
    keys = ('f0', 'ap', 'sp', 'voiced', 'mc')
    digest = lambda f: hashlib.sha1(b''.join(numpy.ascontiguousarray(getattr(f, k)).tobytes() for k in keys)).hexdigest()
    probe_digest, probes, released, t0 = None, 0, 0, time.time()
    sent = {}
    with dispatch.ChunkDispatcher(ac, sr, [0] * workers, comm='host', depth=2, start_timeout=900, worker_hook=hook) as d:
        def take(res):
            nonlocal probe_digest, probes, released
            for index, f in res:
                assert index == released, 'released out of order: %r after %d' % (index, released - 1)
                released += 1
                kind, n, pad = sent.pop(index)
                assert f.sp.shape == (n - 2 * pad, synth.FFT_BINS) and numpy.isfinite(f.sp).all() and (f.sp > 0).all()
                silent = ~f.mc.any(axis=1)                                   # frames the gate cut: mc, ap, f0 are the all-silent zeros
                assert not f.ap[silent].any() and not f.f0[silent].any() and f.ap[~silent].all()
                if kind == 'probe':
                    probes += 1
                    h = digest(f)
                    if probe_digest is None:
                        probe_digest = h
                    assert h == probe_digest, 'the probe window changed its bits at window %d' % index
        for i in range(n_windows):
            if i % 10 == 0:
                f, kind = probe, 'probe'
            else:
                f, kind = pool[int(rng.integers(len(pool)))], 'pool'
            n = len(f.f0)
            pad = 100 if n == 300 else (int(rng.integers(0, n // 4)) if kind == 'pool' and i % 3 else 0)
            if kind == 'probe':
                pad = 100
            sent[i] = (kind, n, pad)
            d.submit(i, f, discard=(pad, pad), pick=(pad, n - pad, keys) if pad else None)
            take(d.collect())
            if (i + 1) % 500 == 0:
                print('window %5d  %.1f s  %.0f windows/s  released %d  ahead-of-order max %d' % (i + 1, time.time() - t0, (i + 1) / (time.time() - t0), released, d.max_out_of_order), flush=True)
        take(d.drain(timeout=600))
    assert released == n_windows and not sent and probes == (n_windows + 9) // 10
    print('dispatcher soak OK: %d windows (%d probes bit-identical) through %d worker processes in %.1f s' % (n_windows, probes, workers, time.time() - t0))


if __name__ == '__main__':
    main()
