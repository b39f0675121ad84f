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


# ---- model files and feature windows for the measurements that go through the import surface (bench.py --dispatcher) -----------------

def write_model_files(d, name: str = 'SYN-64', out_rate: int = 16000) -> None:
    """The two model directories a user of the reference has on disk (Chainer `save_npz` key layout + config.json + the f0 statistics,
    /root/reference/README.md:22-40), filled with the seeded synthetic weights."""
    import json
    from pathlib import Path
    from .weights import save_npz
    d = Path(d)
    (d1, P1), (d2, P2) = model_params(name)
    save_npz(d / 's1.npz', P1)
    save_npz(d / 's2.npz', P2)
    (d / 's1.json').write_text(json.dumps({
        'dataset': {'acoustic_param': {'sampling_rate': 16000, 'frame_period': 5, 'order': 8, 'alpha': 0.41},
                    'in_features': ['mc'], 'out_features': ['mc']},
        'model': {'in_channels': MC_DIMS, 'out_channels': MC_DIMS, 'generator_base_channels': d1.base, 'generator_extensive_layers': 8}}))
    (d / 's2.json').write_text(json.dumps({
        'dataset': {'param': {'voice_param': {'sample_rate': out_rate}, 'acoustic_feature_param': {'frame_period': 5, 'order': 8}}},
        'model': {'generator_base_channels': d2.base, 'generator_extensive_layers': 8}}))
    numpy.save(str(d / 'in_stat.npy'), {'mean': numpy.log(200.0), 'var': 0.04})
    numpy.save(str(d / 'tg_stat.npy'), {'mean': numpy.log(300.0), 'var': 0.09})


def build_converters(d, out_rate: int = 16000, gpu: int = 0):
    """(`yukarin.AcousticConverter`, `become_yukarin.SuperResolution`) over the files of `write_model_files`, as check.py / run.py build
    them (/root/reference/check.py:46-63)."""
    from pathlib import Path
    from . import compat
    compat.install()
    from become_yukarin import SuperResolution
    from become_yukarin.config.sr_config import create_from_json as create_sr_config
    from yukarin import AcousticConverter
    from yukarin.config import create_from_json as create_config
    from yukarin.f0_converter import F0Converter
    d = Path(d)
    f0c = F0Converter(input_statistics=d / 'in_stat.npy', target_statistics=d / 'tg_stat.npy')
    ac = AcousticConverter(create_config(d / 's1.json'), d / 's1.npz', gpu=gpu, f0_converter=f0c, out_sampling_rate=out_rate)
    sr = SuperResolution(create_sr_config(d / 's2.json'), d / 's2.npz', gpu=gpu)
    return ac, sr


def feature_window(n_frames: int, seed: int, silent_stretch: bool = False):
    """One window as `ConvertStream.fetch` hands it to the stage: a `yukarin.AcousticFeature` with f0 / ap / mc / voiced of n_frames and the raw
    wave (16 kHz, 80 samples per frame) as `.wave` (SURVEY.md 8(d) distributions); silent_stretch: a third of the window below the gate."""
    from . import compat
    compat.install()
    from yukarin import AcousticFeature, Wave
    rng = numpy.random.default_rng(seed)
    wave = (0.1 * rng.normal(size=n_frames * 80)).astype(numpy.float32)
    if silent_stretch:
        wave[(n_frames // 3) * 80:(2 * n_frames // 3) * 80] = 0.0
    f0 = numpy.where(rng.random((n_frames, 1)) < 0.3, 0.0, rng.lognormal(numpy.log(220.0), 0.2, (n_frames, 1))).astype(numpy.float32)
    f = AcousticFeature(f0=f0, ap=rng.uniform(0.001, 0.999, (n_frames, FFT_BINS)).astype(numpy.float32),
                        mc=(rng.normal(size=(n_frames, MC_DIMS)) * MC_SCALE).astype(numpy.float32), voiced=f0 > 0)
    f.wave = Wave(wave=wave, sampling_rate=16000)
    return f
