import os
import sys
from pathlib import Path

import numpy
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run by the driver with -m gpu)')


@pytest.fixture(scope='session')
def emu_ctx():
    """Context over the host-side SIMT emulator build of the kernel sources (CPU, test-only)."""
    from realtime_yukarin_amd import _lib, build, engine
    lib = _lib.Ry355Lib(os.environ.get('RY_EMU_LIB') or build.build_emu())      # RY_EMU_LIB: another build of the emulator library (scripts/asan_emu.sh: the sanitizer build)
    return engine.Context(0, lib)


@pytest.fixture(scope='session')
def gpu_ctx():
    """Context over the product library on GPU 0.  Fails (never skips) when the HIP path is unavailable."""
    from realtime_yukarin_amd import _lib, engine
    lib = _lib.default_lib()
    assert lib.device_count() >= 1, 'no HIP device visible'
    return engine.get_context(0)


def rel_max(a, b):
    """max |a - b| / max |b|  (the parity metric for activations: tolerance 1e-4, BASELINE.json north_star)."""
    a = numpy.asarray(a, dtype=numpy.float64)
    b = numpy.asarray(b, dtype=numpy.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(numpy.abs(a - b).max() / max(numpy.abs(b).max(), 1e-30))


def bn_params(rng, c):
    return (rng.normal(1, 0.1, c).astype('f4'), rng.normal(0, 0.1, c).astype('f4'),
            rng.normal(0, 0.1, c).astype('f4'), rng.uniform(0.5, 1.5, c).astype('f4'))
