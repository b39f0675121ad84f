"""Host side of the device silence gate (`ry_vc_submit_wave`).

The gate of `Wave.get_effective_frame` is `10 * log10(max(1e-10, mse)) > -threshold_db` after librosa's `top_db = 80` clamp
(compat/yukarin/wave.py).  Both conditions are monotone in the frame power, so they are evaluated on the device as comparisons with
two float32 thresholds:

    effective[t] = mse[t] >= p_effective   or   max(mse) >= p_all

and the thresholds are found HERE, once per threshold_db, by bisection over the float32 values through the host's own numpy
arithmetic -- the exact expressions of the shim.  Whatever this numpy's float32 log10 returns in its last bit, the device mask agrees
with the host mask bit for bit (checked against oracle/effective_frame.py in tests/test_device_gate.py)."""
import functools

import numpy

from .compat.yukarin import wave as _wave          # the shim's constants and expressions (AMIN, TOP_DB, power_to_db)


def _f32(bits: int) -> numpy.float32:
    return numpy.array([bits], dtype=numpy.uint32).view(numpy.float32)[0]


def _smallest_true(pred) -> numpy.float32:
    """Smallest non-negative float32 p with pred(p), for a monotone predicate; +inf when it is never true."""
    lo, hi = 0, 0x7f800000                               # bit patterns of +0.0 .. +inf order like the values
    if pred(_f32(lo)):
        return _f32(lo)
    if not pred(_f32(hi - 1)):
        return numpy.float32(numpy.inf)
    hi -= 1
    while hi - lo > 1:                                   # invariant: not pred(lo), pred(hi)
        mid = (lo + hi) // 2
        if pred(_f32(mid)):
            hi = mid
        else:
            lo = mid
    return _f32(hi)


@functools.lru_cache(maxsize=64)
def thresholds(threshold_db: float):
    """(p_effective, p_all) for the absolute gate (`ref = 'abs'`)."""
    thr = float(threshold_db)

    def db_of(p):                                        # the shim's expression on a vector (64 equal lanes: numpy's SIMD loop, as for real windows)
        s = numpy.full(64, p, dtype=numpy.float32)
        return 10.0 * numpy.log10(numpy.maximum(_wave.AMIN, s))

    p_eff = _smallest_true(lambda p: bool((db_of(p) > -thr)[0]))
    p_all = _smallest_true(lambda p: bool(((db_of(p) - _wave.TOP_DB) > -thr)[0]))
    return float(p_eff), float(p_all)


def device_gate_usable(wave: numpy.ndarray, fft_length: int, threshold_db, ref: str) -> bool:
    """The device gate restates the float32 arithmetic of the absolute gate for power-of-two frame lengths 128 .. 1024."""
    return (threshold_db is not None and ref == 'abs' and isinstance(wave, numpy.ndarray) and wave.dtype == numpy.float32 and wave.ndim == 1
            and len(wave) > 0 and 128 <= int(fft_length) <= 1024 and (int(fft_length) & (int(fft_length) - 1)) == 0)
