"""Seeded synthetic inputs and the canonical model configs used by tests and bench.

The reference ships neither trained models nor their config.json (/root/reference/README.md:22-40),
so the canonical configs are declared here: SYN-64 (generator_base_channels 64, extensive_layers 8,
stage-1 in/out = 9 mel-cepstrum dims, stage-2 1 -> 1 channel on 512 bins) and SYN-8 (base 8).
"""
import numpy

from .netspec import NetDesc
from .weights import synthetic_params

SEED_INPUT, SEED_STAGE1, SEED_STAGE2 = 355, 356, 357
FFT_BINS = 513          # fft_size / 2 + 1 for 16 kHz / 24 kHz (cheaptrick fft_size 1024)
MC_DIMS = 9             # order 8 -> 9 mel-cepstrum coefficients
MC_SCALE = numpy.array([4, 1, .5, .5, .3, .3, .2, .2, .2], dtype=numpy.float64)


def model_descs(name: str = 'SYN-64', stage1_in: int = MC_DIMS):
    base = {'SYN-64': 64, 'SYN-32': 32, 'SYN-8': 8}[name]
    return NetDesc(1, stage1_in, MC_DIMS, base, 8), NetDesc(2, 1, 1, base, 8)


def model_params(name: str = 'SYN-64', stage1_in: int = MC_DIMS):
    d1, d2 = model_descs(name, stage1_in)
    return (d1, synthetic_params(d1, SEED_STAGE1)), (d2, synthetic_params(d2, SEED_STAGE2))


def stage1_input(n_frames: int, windows: int = 1, seed: int = SEED_INPUT, stress: bool = False) -> numpy.ndarray:
    """(windows, N, C_in): mc ~ N(0,1) * per-coefficient scale; stress variant appends f0 and ap (C_in = 523)."""
    rng = numpy.random.default_rng(seed)
    mc = rng.normal(size=(windows, n_frames, MC_DIMS)) * MC_SCALE
    if not stress:
        return mc.astype(numpy.float32)
    f0 = numpy.where(rng.random((windows, n_frames, 1)) < 0.3, 0.0, rng.lognormal(numpy.log(220.0), 0.2, (windows, n_frames, 1)))
    ap = rng.uniform(0.001, 0.999, (windows, n_frames, FFT_BINS))
    return numpy.concatenate([mc, f0, ap], axis=2).astype(numpy.float32)


def stage2_input(n_frames: int, windows: int = 1, seed: int = SEED_INPUT + 1, bins: int = FFT_BINS) -> numpy.ndarray:
    """(windows, N, bins) spectrogram: exp(N(-6, 1.5)) + 1e-16 (the floor voice_changer.py:39 adds)."""
    rng = numpy.random.default_rng(seed)
    return (numpy.exp(rng.normal(-6.0, 1.5, (windows, n_frames, bins))) + 1e-16).astype(numpy.float32)
