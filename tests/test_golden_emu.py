"""The kernel sources (emulator build, CPU) against the committed golden fixtures -- independent of the oracle code."""
import glob
from pathlib import Path

import numpy
import pytest

from realtime_yukarin_amd import engine
from realtime_yukarin_amd.netspec import NetDesc
from realtime_yukarin_amd.weights import flatten_params, synthetic_params

GOLD = Path(__file__).resolve().parent / 'golden'


def run_fixture(ctx, path):
    z = numpy.load(path)
    nd, inc, outc, base, e = [int(v) for v in z['desc']]
    d = NetDesc(nd, inc, outc, base, e)
    P = synthetic_params(d, int(z['seed']), bias_std=float(z['bias_std']))
    net = engine.Net(ctx, d, flatten_params(d, P), width=int(z['width']) if 'width' in z.files else 512)
    x, y = z['x'], z['y']
    if nd == 2 and 'forward' in Path(path).stem:
        got = net.forward(x)
        err = numpy.abs(got - y).max() / numpy.abs(y).max()
    else:
        got = net.convert(x)
        err = numpy.abs(got / y - 1).max() if nd == 2 else numpy.abs(got - y).max() / numpy.abs(y).max()
    net.close()
    return float(err)


@pytest.mark.parametrize('path', sorted(glob.glob(str(GOLD / '*.npz'))), ids=lambda p: Path(p).stem)
def test_emulated_kernels_match_golden(emu_ctx, path):
    assert run_fixture(emu_ctx, path) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize('path', sorted(glob.glob(str(GOLD / '*.npz'))), ids=lambda p: Path(p).stem)
def test_gpu_matches_golden(gpu_ctx, path):
    assert run_fixture(gpu_ctx, path) < 1e-4
