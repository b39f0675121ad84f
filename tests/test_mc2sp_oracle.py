"""`decode_spectrogram` = pysptk.mc2sp (/root/reference/realtime_voice_conversion/yukarin_wrapper/voice_changer.py:38): the independent
oracle `oracle/mc2sp.py` pinned by hand-computable answers and by its two routes (SPTK call sequence in scalar loops; the all-pass
definition) agreeing, then the PRODUCT's host forms (`realtime_yukarin_amd/sptk.py`: recursion, matrix, mcepalpha) held to it.  The
oracle shares no code with the product."""
import math

import numpy
import pytest

from oracle import mc2sp as omc
from realtime_yukarin_amd import sptk


def test_oracle_imports_nothing_of_the_product():
    import inspect
    src = inspect.getsource(omc)
    assert 'realtime_yukarin_amd' not in src.split('"""', 2)[2]


@pytest.mark.parametrize('fs,alpha', [(16000, 0.41), (24000, 0.466), (44100, 0.544), (48000, 0.554)])
def test_mcepalpha_known_answers(fs, alpha):
    """The published all-pass constants (0.466 is also `AcousticParam.alpha`'s default for its 24 kHz default rate)."""
    assert round(omc.mcepalpha(fs), 3) == alpha
    assert round(sptk.mcepalpha(fs), 3) == alpha                          # the product's atan(num / den) form


def test_hand_computable_spectra():
    for alpha in (0.41, 0.466):
        for route in (omc.mc2sp_sptk, omc.mc2sp_closed):
            c0 = numpy.zeros((1, 9)); c0[0, 0] = -3.0
            assert numpy.allclose(route(c0, alpha, 1024), math.exp(-6.0), rtol=1e-12)           # a flat spectrum: exp(2 c0)
            c1 = numpy.zeros((1, 9)); c1[0, 1] = 1.0
            sp = route(c1, alpha, 1024)[0]
            assert math.isclose(sp[0], math.exp(2.0), rel_tol=1e-12)                            # w = 0 -> w~ = 0
            assert math.isclose(sp[-1], math.exp(-2.0), rel_tol=1e-12)                          # w = pi -> w~ = pi
            k = 100                                                                             # an interior bin, by the textbook phase
            w = 2 * math.pi * k / 1024
            wt = math.atan2((1 - alpha ** 2) * math.sin(w), (1 + alpha ** 2) * math.cos(w) - 2 * alpha)
            assert math.isclose(sp[k], math.exp(2.0 * math.cos(wt)), rel_tol=1e-11)
    # alpha = 0: no warping, the mel-cepstrum IS the cepstrum
    mc = numpy.random.default_rng(1).normal(size=(3, 9))
    w = 2 * numpy.pi * numpy.arange(129) / 256
    plain = numpy.exp(2.0 * mc @ numpy.cos(numpy.outer(numpy.arange(9), w)))
    assert numpy.allclose(omc.mc2sp_sptk(mc, 0.0, 256), plain, rtol=1e-12)


@pytest.mark.parametrize('alpha,fftlen,order', [(0.41, 1024, 8), (0.466, 1024, 8), (0.544, 2048, 24), (0.41, 256, 8)])
def test_the_two_routes_agree(alpha, fftlen, order):
    rng = numpy.random.default_rng(5)
    mc = rng.normal(size=(4, order + 1)) * numpy.linspace(3.0, 0.2, order + 1)
    a, b = omc.mc2sp_sptk(mc, alpha, fftlen), omc.mc2sp_closed(mc, alpha, fftlen)
    assert a.shape == (4, fftlen // 2 + 1)
    assert float(numpy.abs(numpy.log(a) - numpy.log(b)).max()) < 1e-10      # float64 rounding of a 1024-term cosine sum on log-spectra of size ~ 40


@pytest.mark.parametrize('alpha', [0.41, 0.466])
def test_product_host_forms_against_the_oracle(alpha):
    """`sptk.mc2sp` (vectorised recursion + rfft), `sptk.mc2sp_fast` / `mc2sp_matrix` (what the device kernel multiplies by)."""
    rng = numpy.random.default_rng(6)
    mc = rng.normal(size=(50, 9)) * numpy.array([4, 1, .5, .5, .3, .3, .2, .2, .2])
    ref = omc.mc2sp_sptk(mc, alpha, 1024)
    assert float(numpy.abs(sptk.mc2sp(mc, alpha, 1024) / ref - 1).max()) < 1e-12
    assert float(numpy.abs(sptk.mc2sp_fast(mc, alpha, 1024) / ref - 1).max()) < 1e-12
    M = sptk.mc2sp_matrix(8, alpha, 1024)
    w = omc.allpass_phase(2 * numpy.pi * numpy.arange(513) / 1024, alpha)
    assert float(numpy.abs(M - 2.0 * numpy.cos(numpy.outer(numpy.arange(9), w))).max()) < 1e-12
