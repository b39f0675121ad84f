"""Host-side forms of the two SPTK routines `AcousticConverter.decode_spectrogram` uses ([MEM]:
`pysptk.mc2sp(mc, alpha=pysptk.util.mcepalpha(out_rate), fftlen=1024)`, reached from
/root/reference/realtime_voice_conversion/yukarin_wrapper/voice_changer.py:38).  In the reference this runs on the CPU in SPTK C
between the two CNNs.  Here the window call forms the spectrogram on the device (`ry_mc2sp`: exp(mc @ M), M = `mc2sp_matrix`
below, built once per converter); the host recursion `mc2sp` / the host matrix form `mc2sp_fast` remain for callers that read the
intermediate spectrogram on the host.  The checker of all three is `oracle/mc2sp.py`, which shares no code with this file."""
import numpy


def mcepalpha(fs: int, start: float = 0.0, stop: float = 1.0, step: float = 0.001, num_points: int = 1000) -> float:
    """pysptk.util.mcepalpha: all-pass constant whose warping is closest (RMS) to the mel scale."""
    alphas = numpy.arange(start, stop, step)
    hz = (fs / 2.0) / num_points * numpy.arange(num_points)
    mel = 1000.0 / numpy.log(2) * numpy.log(1 + hz / 1000.0)
    mel = mel / mel[-1]
    omega = numpy.pi / num_points * numpy.arange(num_points)
    best, best_d = alphas[0], numpy.inf
    for a in alphas:
        warp = numpy.arctan2((1 - a * a) * numpy.sin(omega), (1 + a * a) * numpy.cos(omega) - 2 * a)
        warp[warp < 0] += numpy.pi
        warp = warp / warp[-1]
        d = numpy.sqrt(numpy.mean((mel - warp) ** 2))
        if d < best_d:
            best, best_d = a, d
    return float(best)


def freqt(c: numpy.ndarray, order: int, alpha: float) -> numpy.ndarray:
    """SPTK freqt (frequency transform of cepstra), vectorised over frames: c (N, m1+1) -> (N, order+1)."""
    c = numpy.asarray(c, dtype=numpy.float64)
    n, m1 = c.shape[0], c.shape[1] - 1
    beta = 1.0 - alpha * alpha
    g = numpy.zeros((n, order + 1))
    for i in range(-m1, 1):
        d = g.copy()
        g[:, 0] = c[:, -i] + alpha * d[:, 0]
        if order >= 1:
            g[:, 1] = beta * d[:, 0] + alpha * d[:, 1]
        for j in range(2, order + 1):
            g[:, j] = d[:, j - 1] + alpha * (d[:, j] - g[:, j - 1])
    return g


def mc2sp(mc: numpy.ndarray, alpha: float, fftlen: int) -> numpy.ndarray:
    """pysptk.mc2sp: mel-cepstrum (N, order+1) -> power spectrum (N, fftlen/2+1)."""
    c = freqt(mc, fftlen // 2, -alpha)
    c[:, 0] *= 2.0
    symc = numpy.zeros((c.shape[0], fftlen))
    symc[:, 0] = c[:, 0]
    symc[:, 1:c.shape[1]] = c[:, 1:]
    symc[:, -1:-c.shape[1]:-1] = c[:, 1:]
    return numpy.exp(numpy.fft.rfft(symc, axis=1).real)


_MATRIX_CACHE = {}


def mc2sp_matrix(order: int, alpha: float, fftlen: int) -> numpy.ndarray:
    """(order+1, fftlen/2+1) float64 matrix M with mc2sp(mc) == exp(mc @ M).

    Every step of mc2sp before the exp is linear in the mel-cepstrum (freqt is a linear recursion, then c0 *= 2, the
    symmetric extension and the real part of the rfft), so M is mc2sp's linear part applied to the identity.  This is
    what lets `decode_spectrogram` run on the GPU between the two CNNs as one small matmul + exp (`ry_mc2sp`)."""
    key = (int(order), float(alpha), int(fftlen))
    m = _MATRIX_CACHE.get(key)
    if m is None:
        c = freqt(numpy.eye(order + 1), fftlen // 2, -alpha)
        c[:, 0] *= 2.0
        symc = numpy.zeros((order + 1, fftlen))
        symc[:, 0] = c[:, 0]
        symc[:, 1:c.shape[1]] = c[:, 1:]
        symc[:, -1:-c.shape[1]:-1] = c[:, 1:]
        m = numpy.fft.rfft(symc, axis=1).real
        _MATRIX_CACHE[key] = m
    return m


def mc2sp_fast(mc: numpy.ndarray, alpha: float, fftlen: int) -> numpy.ndarray:
    """Host form of the same identity (float64): exp(mc @ M); ~100x cheaper than the recursion per window."""
    mc = numpy.asarray(mc, dtype=numpy.float64)
    return numpy.exp(mc @ mc2sp_matrix(mc.shape[1] - 1, alpha, fftlen))
