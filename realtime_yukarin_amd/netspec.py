"""Network description of the two predictors on the convert hot path (host side).

The reference builds both predictors inside un-vendored dependencies
(`yukarin.AcousticConverter.__init__` / `become_yukarin.SuperResolution.__init__`, constructed at
/root/reference/realtime_voice_conversion/converter/yukarin_converter.py:40-55 and
/root/reference/check.py:54-63) from the `model` section of each stage's config.json.  This module
restates the topology as data: an ordered K-list of
Chainer `save_npz` keys with shapes.  The flat weight blob handed to the C-ABI
(`ry_net_create`, include/ry355.h) is the concatenation of these arrays in exactly this order;
`csrc/ry_plan.cpp` derives the same order from the same five integers and both sides cross-check
the element count.
"""
from dataclasses import dataclass
from typing import List, Tuple

import numpy

ENC_CH = (1, 2, 4, 8, 8, 8, 8, 8)
DEC_IN = (8, 16, 16, 16, 16, 8, 4)
DEC_OUT = (8, 8, 8, 8, 4, 2, 1)
BN_KEYS = ('gamma', 'beta', 'avg_mean', 'avg_var')


@dataclass(frozen=True)
class NetDesc:
    """ndim 1 = stage-1 1-D U-Net (Convolution1D), ndim 2 = stage-2 2-D U-Net (Convolution2D)."""
    ndim: int
    in_ch: int
    out_ch: int
    base: int = 64
    extensive_layers: int = 8
    glu: bool = False          # stage-1 `model.glu_generator` (UNVERIFIED [MEM]): gated blocks, twice the conv / BN channels per block

    def __post_init__(self):
        if self.ndim not in (1, 2):
            raise ValueError('ndim must be 1 or 2')
        if self.glu and self.ndim != 1:
            raise ValueError('glu_generator is a stage-1 option')
        if min(self.in_ch, self.out_ch, self.base) < 1 or not (0 <= self.extensive_layers <= 8):
            raise ValueError('bad NetDesc %r' % (self,))


def _k(desc: NetDesc, size: int) -> Tuple[int, ...]:
    return (size,) * desc.ndim


def enc_sample(desc: NetDesc, i: int) -> str:
    return 'down' if i < desc.extensive_layers else 'same'


def dec_sample(desc: NetDesc, j: int) -> str:
    return 'up' if (7 - j) < desc.extensive_layers else 'same'


def param_list(desc: NetDesc) -> List[Tuple[str, Tuple[int, ...]]]:
    """Ordered (key, shape) for every float array of the predictor (BN counter `N` excluded)."""
    B = desc.base
    g = 2 if desc.glu else 1          # a gated block computes 2 x co channels (value | gate) and hands co on
    end_k = 3 if desc.extensive_layers > 0 else 1
    out = [('encoder/c0/W', (B, desc.in_ch) + _k(desc, end_k)), ('encoder/c0/b', (B,))]
    for i in range(1, 8):
        ci, co = ENC_CH[i - 1] * B, ENC_CH[i] * B * g
        k = 4 if enc_sample(desc, i) == 'down' else 1
        p = 'encoder/c%d' % i
        out += [(p + '/c/W', (co, ci) + _k(desc, k)), (p + '/c/b', (co,))]
        out += [(p + '/batchnorm/' + n, (co,)) for n in BN_KEYS]
    for j in range(0, 7):
        ci, co = DEC_IN[j] * B, DEC_OUT[j] * B * g
        p = 'decoder/c%d' % j
        if dec_sample(desc, j) == 'up':
            out += [(p + '/c/W', (ci, co) + _k(desc, 4))]          # Deconvolution: (Cin, Cout, k...)
        else:
            out += [(p + '/c/W', (co, ci) + _k(desc, 1))]
        out += [(p + '/c/b', (co,))]
        out += [(p + '/batchnorm/' + n, (co,)) for n in BN_KEYS]
    out += [('decoder/c7/W', (desc.out_ch, 2 * B) + _k(desc, end_k)), ('decoder/c7/b', (desc.out_ch,))]
    return out


def param_count(desc: NetDesc) -> int:
    return int(sum(int(numpy.prod(s)) for _, s in param_list(desc)))


def pad_frames(n: int) -> int:
    """Both wrappers pad time by 128 - n % 128 (a full 128 when n is already a multiple)."""
    return 128 - n % 128


def flops(desc: NetDesc, T: int, width: int = 1) -> int:
    """Algorithmic FLOPs of one forward at padded length T (stage-2: T x width image).
    conv: 2*Cin*Cout*k^n*out_spatial; k4s2 deconv: 2*Cin*Cout*k^n*in_spatial."""
    n = desc.ndim
    sp = [T] if n == 1 else [T, width]
    B = desc.base
    g = 2 if desc.glu else 1
    end_k = 3 if desc.extensive_layers > 0 else 1
    area = lambda s: int(numpy.prod(s))
    total = 2 * desc.in_ch * B * end_k ** n * area(sp)
    cur = list(sp)
    sizes = [list(cur)]
    for i in range(1, 8):
        ci, co = ENC_CH[i - 1] * B, ENC_CH[i] * B * g
        if enc_sample(desc, i) == 'down':
            cur = [c // 2 for c in cur]
            total += 2 * ci * co * 4 ** n * area(cur)
        else:
            total += 2 * ci * co * area(cur)
        sizes.append(list(cur))
    for j in range(0, 7):
        ci, co = DEC_IN[j] * B, DEC_OUT[j] * B * g
        if dec_sample(desc, j) == 'up':
            total += 2 * ci * co * 4 ** n * area(cur)
            cur = [c * 2 for c in cur]
        else:
            total += 2 * ci * co * area(cur)
    total += 2 * 2 * B * desc.out_ch * end_k ** n * area(cur)
    return int(total)
