"""Round-trip time of one feature item parent -> child -> parent: multiprocessing.Queue (what run.py:48-51 builds) against
realtime_yukarin_amd.transport.FeatureQueue.  CPU only.  Usage: python scripts/transport_bench.py [frames ...]"""
import multiprocessing
import sys
from pathlib import Path

import numpy

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from realtime_yukarin_amd import compat                                   # noqa: E402
from realtime_yukarin_amd.transport import FeatureQueue, measure_round_trip   # noqa: E402

compat.install()


class Item(object):
    def __init__(self, item, index):
        self.item = item
        self.index = index


def feature(n, dtype):
    from yukarin import AcousticFeature, Wave
    rng = numpy.random.default_rng(0)
    f = AcousticFeature(f0=rng.random((n, 1)).astype(dtype), ap=rng.random((n, 513)).astype(dtype), sp=rng.random((n, 513)).astype(dtype),
                        mc=rng.normal(size=(n, 9)).astype(dtype), voiced=rng.random((n, 1)) > 0.5)
    f.wave = Wave(wave=rng.normal(size=n * 80).astype(dtype), sampling_rate=16000)
    return f


if __name__ == '__main__':
    frames = [int(a) for a in sys.argv[1:]] or [100, 300, 1000]
    print('%-8s %-8s %10s %14s %14s %8s' % ('frames', 'dtype', 'MB/item', 'mp.Queue ms', 'FeatureQueue ms', 'ratio'))
    for n in frames:
        for dtype in (numpy.float32, numpy.float64):      # convert-side items are fp32, decode-side fp64 (vocoder.py:54)
            item = Item(feature(n, dtype), 0)
            mb = sum(getattr(item.item, k).nbytes for k in ('f0', 'ap', 'sp', 'mc', 'voiced')) / 1e6 + item.item.wave.wave.nbytes / 1e6
            t_pipe = measure_round_trip(multiprocessing.Queue, item, n=60, warmup=5)
            t_shm = measure_round_trip(lambda: FeatureQueue(slots=4, slot_bytes=32 << 20), item, n=60, warmup=5)
            print('%-8d %-8s %10.2f %14.3f %14.3f %8.1fx' % (n, numpy.dtype(dtype).name, mb, t_pipe * 1e3, t_shm * 1e3, t_pipe / t_shm))
