#!/usr/bin/env python3
"""The WHOLE central loop of `dispatch.convert_worker_multi_gpu` -- the reference's own `ConvertStream` (`add` / `fetch` / `remove`,
/root/reference/realtime_voice_conversion/stream/base_stream.py:20-79), the window hand-out, the in-order release -- with null workers
(no GPU, no arithmetic: they answer at once, or after `pace` ms of busy waiting as a stand-in for one GPU's time per window), items in
and out through `transport.FeatureQueue` as run.py's queues would carry them.  Says how many 0.5 s buffers per second ONE dispatcher
process can fetch, hand out and release, i.e. how many GPUs it can feed.  Needs /root/reference (runs in the authoring container).

usage: python scripts/dispatch_loop_bench.py [workers] [items] [pace_ms] [prof]      (prof: cProfile of the loop)"""
import sys
import threading
import time
from pathlib import Path

import numpy

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / 'tests' / 'stubs'), '/root/reference'):
    sys.path.insert(0, p)


def _feeder(q_in, n_items, go):
    """Its own process (no GIL shared with the loop under test): builds nine 100-frame buffers and sends n_items of them."""
    for p in (str(ROOT), str(ROOT / 'tests' / 'stubs'), '/root/reference'):
        sys.path.insert(0, p)
    from realtime_yukarin_amd import compat, synth
    compat.install()
    from realtime_voice_conversion.worker.utility import Item
    from realtime_voice_conversion.yukarin_wrapper.voice_changer import AcousticFeatureWrapper
    feats = []
    for i in range(9):                                      # one 0.5 s buffer = 100 frames: what the encode worker sends per item
        f = synth.feature_window(100, 300 + i)
        feats.append(AcousticFeatureWrapper(wave=f.wave, f0=f.f0, ap=f.ap, mc=f.mc, voiced=f.voiced))
    go.wait()
    for i in range(n_items):
        q_in.put(Item(item=feats[i % 9], index=i))
    q_in.put(None)


def _drainer(q_out, n_items, done):
    for p in (str(ROOT), str(ROOT / 'tests' / 'stubs'), '/root/reference'):
        sys.path.insert(0, p)
    from realtime_yukarin_amd import compat
    compat.install()
    last = -1
    for _ in range(n_items):
        it = q_out.get(timeout=300)
        assert it.index == last + 1, 'out of order'
        last = it.index
    done.set()


def main():
    import multiprocessing
    import tempfile
    from realtime_yukarin_amd import compat, dispatch, synth
    from realtime_yukarin_amd.transport import FeatureQueue
    compat.install()
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n_items = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    pace = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    d = Path(tempfile.mkdtemp()); synth.write_model_files(d, 'SYN-8'); ac, sr = synth.build_converters(d)
    ctx = multiprocessing.get_context('spawn')
    q_in, q_out = FeatureQueue(slots=32, slot_bytes=2 << 20, ctx=ctx), FeatureQueue(slots=32, slot_bytes=2 << 20, ctx=ctx)
    go, done = ctx.Event(), ctx.Event()
    pf = ctx.Process(target=_feeder, args=(q_in, n_items, go), daemon=True); pf.start()
    pd = ctx.Process(target=_drainer, args=(q_out, n_items, done), daemon=True); pd.start()
    lock = threading.Lock(); lock.acquire()
    kw = dict(devices=[0] * G, comm='host', null_workers=True, worker_hook=dispatch.paced(pace) if pace > 0 else None)
    t0 = [0.0]

    def starter():
        lock.acquire()                                      # the workers are up: the feeder may start, the clock runs
        time.sleep(1.0)
        t0[0] = time.perf_counter(); go.set()
    threading.Thread(target=starter, daemon=True).start()
    if len(sys.argv) > 4:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
    dispatch.convert_worker_multi_gpu(ac, sr, 0.5, 0.5, 60, q_in, q_out, lock, **kw)     # in THIS thread: the loop under test
    if len(sys.argv) > 4:
        pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
    done.wait(60)
    el = time.perf_counter() - t0[0]
    print('%d null workers (pace %.2f ms): %d items in %.3f s = %.0f buffers/s through convert_worker_multi_gpu (300-frame windows fetched by the '
          'reference ConvertStream, 100 kept; feeder and consumer in processes of their own) on %d host cpus' % (G, pace, n_items, el, n_items / el, __import__('os').cpu_count()))


if __name__ == '__main__':
    main()
