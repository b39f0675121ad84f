"""Loop-per-frame restatement of the silence gate of `VoiceChanger.convert_from_acoustic_feature`
(/root/reference/realtime_voice_conversion/yukarin_wrapper/voice_changer.py:27-31 -> `AcousticConverter.separate_effective` ->
`Wave.get_effective_frame`, bodies [MEM]):

    mse = librosa.feature.rms(y, frame_length=fft_length, hop_length=hop, center=True, pad_mode='reflect') ** 2
    effective = librosa.power_to_db(mse.squeeze()) > -threshold_db

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Independent of realtime_yukarin_amd/compat/yukarin/wave.py: no numpy.pad, no
strided frame view, no vectorised mean -- the reflect padding is index arithmetic, every frame is a Python loop, and the frame sum is
written out in the order numpy's float reduction uses for the layout librosa hands it (`abs(x) ** 2` of the strided frame view keeps
the frame axis contiguous, so `mean(axis=0)` is numpy's PAIRWISE summation per frame; pinned against numpy 2.2 in
tests/test_effective_frame.py::test_pairwise_model_is_numpys).  The mask is index / boolean work: the tests demand bit-exact
equality with the shim.

PARITY UNPINNED like the rest of the oracle: librosa cannot be installed here.  `ref` = 'abs' (ref=1.0, top_db=80: a plain
`power_to_db(mse)` call) or 'max' (ref=numpy.max, top_db=None: `librosa.effects.split` style); compat/yukarin/wave.py says which is the
default and why."""
import numpy

AMIN = 1e-10
TOP_DB = 80.0


def reflect_index(i: int, n: int) -> int:
    """Source index of position i (may be negative or >= n) under numpy.pad(mode='reflect') of a length-n axis."""
    if n == 1:
        return 0
    period = 2 * (n - 1)
    j = i % period
    return j if j < n else period - j


def pairwise_sum(a, lo: int, n: int, dtype):
    """numpy's pairwise float summation (umath loops_utils `pairwise_sum`) of a[lo:lo+n], every add rounded to `dtype`."""
    if n < 8:
        r = dtype(0.0)
        for i in range(n):
            r = dtype(r + a[lo + i])
        return r
    if n <= 128:
        r = [dtype(a[lo + j]) for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] = dtype(r[j] + a[lo + i + j])
            i += 8
        res = dtype(dtype(dtype(r[0] + r[1]) + dtype(r[2] + r[3])) + dtype(dtype(r[4] + r[5]) + dtype(r[6] + r[7])))
        while i < n:
            res = dtype(res + a[lo + i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return dtype(pairwise_sum(a, lo, n2, dtype) + pairwise_sum(a, lo + n2, n - n2, dtype))


def frame_power(wave, fft_length: int, hop: int) -> numpy.ndarray:
    y = numpy.asarray(wave)
    if y.dtype.kind != 'f':
        y = y.astype(numpy.float32)
    dtype = y.dtype.type
    n = len(y)
    half = fft_length // 2
    n_frames = 1 + n // hop                       # 1 + (n + 2 * half - fft_length) // hop for an even fft_length
    if fft_length % 2:
        n_frames = 1 + (n + 2 * half - fft_length) // hop
    out = numpy.zeros(n_frames, dtype=y.dtype)
    for t in range(n_frames):
        sq = []
        for k in range(fft_length):
            v = y[reflect_index(t * hop + k - half, n)]
            sq.append(dtype(abs(v) * abs(v)))
        mean = dtype(pairwise_sum(sq, 0, fft_length, dtype) / dtype(fft_length))
        rms = dtype(numpy.sqrt(mean))
        out[t] = dtype(rms * rms)
    return out


def effective_frames(wave, sampling_rate: int, threshold_db: float, fft_length: int, frame_period: float, ref: str = 'abs') -> numpy.ndarray:
    """(len(wave) // hop + 1,) bool."""
    hop = int(sampling_rate * frame_period // 1000)
    n = len(wave)
    if n == 0:
        return numpy.zeros(0, dtype=bool)
    p = frame_power(wave, fft_length, hop)
    db = 10.0 * numpy.log10(numpy.maximum(AMIN, p))
    if ref == 'max':
        db = db - 10.0 * numpy.log10(numpy.maximum(AMIN, p.max()))
    else:
        db = numpy.maximum(db, db.max() - TOP_DB)
    mask = db > -threshold_db
    length = n // hop + 1
    out = numpy.zeros(length, dtype=bool)
    m = min(length, len(mask))
    out[:m] = mask[:m]
    return out


def separate_effective_mask(wave, sampling_rate: int, n_feature_frames: int, threshold_db, fft_length: int, frame_period: float,
                            ref: str = 'abs') -> numpy.ndarray:
    """The mask `AcousticConverter.separate_effective` indexes the feature with: the wave's mask cut / zero-extended to the feature's
    frame count; all-True when the threshold is None."""
    if threshold_db is None:
        return numpy.ones(n_feature_frames, dtype=bool)
    m = effective_frames(wave, sampling_rate, threshold_db, fft_length, frame_period, ref)
    out = numpy.zeros(n_feature_frames, dtype=bool)
    k = min(n_feature_frames, len(m))
    out[:k] = m[:k]
    return out
