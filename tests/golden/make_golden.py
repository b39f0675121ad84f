#!/usr/bin/env python3
"""Generates tests/golden/*.npz: seeded inputs + the ORACLE's outputs for small canonical cases.

PARITY UNPINNED: the reference's arithmetic for this path lives in packages that cannot be imported here
(`yukarin`, `become_yukarin`, `chainer`; /root/reference/requirements.txt:7-8) and the reference ships no golden
vector for either CNN, so these fixtures are produced by our own restatement (`oracle/`, float64 arithmetic,
stored as float32).  They pin the oracle against regressions and give the GPU tests a committed target that does
not depend on the oracle code at test time.  Run from the repo root: `python tests/golden/make_golden.py`.
"""
import sys
from pathlib import Path

import numpy

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))

from oracle import unet  # noqa: E402
from realtime_yukarin_amd.netspec import NetDesc  # noqa: E402
from realtime_yukarin_amd.weights import synthetic_params  # noqa: E402

OUT = Path(__file__).resolve().parent


def f64(P):
    return {k: v.astype(numpy.float64) for k, v in P.items()}


def main():
    rng = numpy.random.default_rng(20260925)
    # stage-1: SYN-8 and a base-32 predictor, convert() array part at N = 100 and N = 37
    for base, n in ((8, 100), (32, 37)):
        d = NetDesc(1, 9, 9, base, 8)
        P = synthetic_params(d, 356, bias_std=0.05)
        x = (rng.normal(size=(n, 9)) * [4, 1, .5, .5, .3, .3, .2, .2, .2]).astype(numpy.float32)
        y = unet.stage1_convert_core(x.astype(numpy.float64), f64(P))
        numpy.savez_compressed(OUT / ('stage1_base%d_n%d.npz' % (base, n)), x=x, y=y.astype(numpy.float32),
                               desc=numpy.array([1, 9, 9, base, 8]), seed=356, bias_std=0.05)
    # stage-2: SYN-8 on a 128-bin spectrogram (width 128 + 1 dropped bin), N = 50
    d = NetDesc(2, 1, 1, 8, 8)
    P = synthetic_params(d, 357, bias_std=0.05)
    sp = (numpy.exp(rng.normal(-6, 1.5, size=(50, 129))) + 1e-16).astype(numpy.float32)
    y = unet.stage2_convert(sp.astype(numpy.float64), f64(P))
    numpy.savez_compressed(OUT / 'stage2_base8_n50.npz', x=sp, y=y.astype(numpy.float32), desc=numpy.array([2, 1, 1, 8, 8]),
                           seed=357, bias_std=0.05, width=128)
    # stage-2: base 32 (implicit-GEMM middle layers), raw predictor on a 128 x 128 block
    d = NetDesc(2, 1, 1, 32, 8)
    P = synthetic_params(d, 357, bias_std=0.05)
    x = rng.normal(size=(1, 128, 128)).astype(numpy.float32)
    y = unet.unet_forward(x[:, None].astype(numpy.float64), f64(P))[:, 0]
    numpy.savez_compressed(OUT / 'stage2_base32_forward128.npz', x=x, y=y.astype(numpy.float32), desc=numpy.array([2, 1, 1, 32, 8]),
                           seed=357, bias_std=0.05, width=128)
    print('wrote', sorted(p.name for p in OUT.glob('*.npz')))


if __name__ == '__main__':
    main()
