"""Independent restatement of `pysptk.mc2sp` / `pysptk.util.mcepalpha`.  TEST INFRASTRUCTURE ONLY.

What it stands in for: `AcousticConverter.decode_spectrogram` = `pysptk.mc2sp(mc, alpha=pysptk.util.mcepalpha(out_rate),
fftlen=1024)` ([MEM] body; reached from /root/reference/realtime_voice_conversion/yukarin_wrapper/voice_changer.py:38, the rate
24000 hard-coded at /root/reference/realtime_voice_conversion/converter/yukarin_converter.py:46).  pysptk (0.1.x, un-pinned in the
reference: it arrives through `yukarin`'s own requirements) is not installable here, so this is a restatement -- PARITY UNPINNED, like
the rest of `oracle/` -- but it SHARES NOTHING with the product's `realtime_yukarin_amd/sptk.py` (no import, no common helper):

* `mc2sp_sptk`      the published call sequence, one frame at a time with scalar loops: SPTK `freqt.c` (the recursion over the input
                    coefficients with the two work rows `d` / `g`), `c[0] *= 2`, the symmetric extension written entry by entry as
                    pysptk does, and the real part of the DFT as an explicit cosine sum (no FFT routine);
* `mc2sp_closed`    the DEFINITION the recursion approximates: a mel-cepstrum is the cepstrum on the frequency axis warped by the
                    first-order all-pass z~^-1 = (z^-1 - alpha) / (1 - alpha z^-1), so
                        log |H(e^jw)|^2 = 2 * sum_m mc[m] * cos(m * w~(w)),   w~ = w + 2 atan2(alpha sin w, 1 - alpha cos w);
                    `freqt` to order fftlen/2 followed by the DFT truncates this series at 512 terms (alpha^512 ~ 1e-170: nothing);
* `mcepalpha`       the grid search of `pysptk.util.mcepalpha` with the warping written as the all-pass phase above instead of the
                    atan(num / den) form the product uses; the widely published values 0.41 @ 16 kHz, 0.466 @ 24 kHz, 0.544 @ 44.1 kHz,
                    0.554 @ 48 kHz are the known answers (`tests/test_mc2sp_oracle.py`).

Two routes that share no code agreeing to 1e-10 on the log-spectrum (float64 rounding of the 1024-term sums) is the pin this file offers; the product's matrix form exp(mc @ M) and the device
kernel `ry_mc2sp` are checked against THESE, never against `realtime_yukarin_amd.sptk`."""
import math

import numpy


def allpass_phase(omega, alpha):
    """Warped frequency w~(w) of the first-order all-pass with constant alpha (0 <= w <= pi, |alpha| < 1)."""
    omega = numpy.asarray(omega, dtype=numpy.float64)
    return omega + 2.0 * numpy.arctan2(alpha * numpy.sin(omega), 1.0 - alpha * numpy.cos(omega))


def mcepalpha(fs, start=0.0, stop=1.0, step=0.001, num_points=1000):
    """All-pass constant whose warping is closest (RMS over `num_points` frequencies) to the mel scale; both curves normalised by
    their last sample, as pysptk.util.mcepalpha does."""
    hz = [(fs / 2.0) / num_points * i for i in range(num_points)]
    mel = numpy.array([1000.0 / math.log(2.0) * math.log(1.0 + f / 1000.0) for f in hz])
    mel /= mel[-1]
    omega = numpy.array([math.pi / num_points * i for i in range(num_points)])
    best, best_d = None, float('inf')
    n_alpha = int(math.ceil((stop - start) / step))
    for k in range(n_alpha):
        a = start + k * step
        warp = allpass_phase(omega, a)
        warp = warp / warp[-1]
        d = math.sqrt(float(numpy.mean((mel - warp) ** 2)))
        if d < best_d:
            best, best_d = a, d
    return float(best)


def freqt_frame(c1, m2, a):
    """SPTK freqt.c for one frame, scalar loops: c1 (m1 + 1 coefficients) -> m2 + 1 coefficients, all-pass constant a."""
    m1 = len(c1) - 1
    b = 1.0 - a * a
    d = [0.0] * (m2 + 1)
    g = [0.0] * (m2 + 1)
    for i in range(-m1, 1):
        d[0] = g[0]
        g[0] = c1[-i] + a * d[0]
        if m2 >= 1:
            d[1] = g[1]
            g[1] = b * d[0] + a * d[1]
        for j in range(2, m2 + 1):
            d[j] = g[j]
            g[j] = d[j - 1] + a * (d[j] - g[j - 1])
    return g


_COS = {}


def _cos_table(fftlen):
    t = _COS.get(fftlen)
    if t is None:
        n = numpy.arange(fftlen, dtype=numpy.float64)
        k = numpy.arange(fftlen // 2 + 1, dtype=numpy.float64)
        t = numpy.cos(2.0 * numpy.pi * numpy.outer(n, k) / fftlen)          # Re of the DFT kernel, (fftlen, fftlen/2 + 1)
        _COS[fftlen] = t
    return t


def mc2sp_sptk(mc, alpha, fftlen):
    """pysptk.mc2sp frame by frame: freqt(mc, fftlen / 2, -alpha), c0 doubled, symmetric extension, exp(Re DFT).  (N, M) -> (N, fftlen/2+1)."""
    mc = numpy.asarray(mc, dtype=numpy.float64)
    cos_t = _cos_table(fftlen)
    out = numpy.empty((mc.shape[0], fftlen // 2 + 1))
    for f in range(mc.shape[0]):
        c = freqt_frame([float(v) for v in mc[f]], fftlen // 2, -alpha)
        c[0] *= 2.0
        symc = [0.0] * fftlen
        symc[0] = c[0]
        for i in range(1, len(c)):
            symc[i] = c[i]
            symc[-i] = c[i]
        out[f] = numpy.exp(numpy.asarray(symc) @ cos_t)
    return out


def mc2sp_closed(mc, alpha, fftlen):
    """The definition: exp(2 sum_m mc[m] cos(m w~(w_k))), w_k = 2 pi k / fftlen, k = 0 .. fftlen/2."""
    mc = numpy.asarray(mc, dtype=numpy.float64)
    w = 2.0 * numpy.pi * numpy.arange(fftlen // 2 + 1) / fftlen
    wt = allpass_phase(w, alpha)
    basis = numpy.cos(numpy.outer(numpy.arange(mc.shape[1]), wt))           # (M, bins)
    return numpy.exp(2.0 * (mc @ basis))


def mc2sp(mc, alpha, fftlen):
    """What the tests call: the closed form (vectorised, cheap at any window); `tests/test_mc2sp_oracle.py` holds it to the SPTK call
    sequence above."""
    return mc2sp_closed(mc, alpha, fftlen)
