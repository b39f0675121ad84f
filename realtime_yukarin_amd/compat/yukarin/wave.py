"""`yukarin.wave.Wave` ([MEM]; used at /root/reference/realtime_voice_conversion/yukarin_wrapper/
acoustic_feature_wrapper.py:23,45,52 and stream/encode_stream.py:35; `get_effective_frame` is what
`AcousticConverter.separate_effective` calls for /root/reference/realtime_voice_conversion/yukarin_wrapper/voice_changer.py:27-31).

`get_effective_frame` restates the librosa call sequence the upstream method is built from ([MEM], unverified here -- librosa is not
installable):

    mse = librosa.feature.rms(y=wave, frame_length=fft_length, hop_length=hop, center=True, pad_mode='reflect') ** 2
    effective = librosa.power_to_db(mse.squeeze()) > -threshold_db                       # ref = 1.0, amin = 1e-10, top_db = 80

with the same numpy primitives in the same order and dtype (so that, on the same numpy, the mask is the one librosa gives):
reflect-pad by fft_length // 2, frames as a strided (fft_length, n_frames) view, `mean(abs(x) ** 2, axis=0)`, `sqrt`, `** 2`,
`10 * log10(maximum(amin, .))`, the `top_db` clamp, the comparison.

Which reference level?  The gate is ABSOLUTE (`ref = 1.0`) by default: (i) the reference has an `else` branch for "no effective
frame" (voice_changer.py:32-35) that a relative gate can never take (its loudest frame is always 0 dB); (ii) the same author gates
the OUTPUT with an absolute `librosa.core.power_to_db(...)` (worker/decode_worker.py:57-58); (iii) the sample config sets both
thresholds to 80 dB (config.yaml:11-12), which as a relative gate would pass everything.  `RY_EFFECTIVE_REF=max` (or
`ref='max'`) selects the relative form (`power_to_db(ref=numpy.max, top_db=None)`, what `librosa.effects.split` does) until a
real install settles it; INTEGRATION.md lists this among the unverified assumptions."""
import os

import numpy

AMIN = 1e-10                    # librosa.power_to_db default
TOP_DB = 80.0                   # librosa.power_to_db default


def default_effective_ref() -> str:
    ref = os.environ.get('RY_EFFECTIVE_REF', 'abs')
    if ref not in ('abs', 'max'):
        raise ValueError("RY_EFFECTIVE_REF must be 'abs' or 'max', not %r" % ref)
    return ref


def frame_power(y: numpy.ndarray, fft_length: int, hop: int) -> numpy.ndarray:
    """`librosa.feature.rms(y, frame_length, hop_length, center=True, pad_mode='reflect') ** 2`, squeezed: (n_frames,)."""
    yp = numpy.pad(y, int(fft_length // 2), mode='reflect')
    n_frames = 1 + (len(yp) - fft_length) // hop
    item = yp.strides[0]
    x = numpy.lib.stride_tricks.as_strided(yp, shape=(fft_length, n_frames), strides=(item, hop * item), writeable=False)   # librosa.util.frame
    power = numpy.mean(numpy.abs(x) ** 2, axis=0, keepdims=True)
    rms = numpy.sqrt(power)
    return (rms ** 2).squeeze(axis=0)


def power_to_db(s: numpy.ndarray, ref: str) -> numpy.ndarray:
    """`librosa.power_to_db(S)` (ref 1.0, top_db 80) or `librosa.power_to_db(S, ref=numpy.max, top_db=None)`."""
    magnitude = numpy.abs(s)
    ref_value = numpy.abs(numpy.max(magnitude)) if ref == 'max' else 1.0
    log_spec = 10.0 * numpy.log10(numpy.maximum(AMIN, magnitude))
    log_spec -= 10.0 * numpy.log10(numpy.maximum(AMIN, ref_value))
    if ref != 'max':
        log_spec = numpy.maximum(log_spec, log_spec.max() - TOP_DB)
    return log_spec


class Wave(object):
    def __init__(self, wave: numpy.ndarray, sampling_rate: int) -> None:
        self.wave = wave
        self.sampling_rate = sampling_rate

    def get_hop_and_length(self, frame_period: float):
        hop = self.sampling_rate * frame_period // 1000
        length = int(len(self.wave) / hop) + 1
        return int(hop), length

    def get_effective_frame(self, threshold_db: float, fft_length: int, frame_period: float, ref: str = None) -> numpy.ndarray:
        """Per-frame bool mask (`length` = len(wave) // hop + 1 frames): frame power in dB above -threshold_db."""
        ref = default_effective_ref() if ref is None else ref
        hop, length = self.get_hop_and_length(frame_period)
        y = numpy.asarray(self.wave)
        if y.dtype.kind != 'f':
            y = y.astype(numpy.float32)
        if len(y) == 0:
            return numpy.zeros(0, dtype=bool)
        effective = power_to_db(frame_power(y, int(fft_length), hop), ref) > -threshold_db
        if len(effective) < length:
            effective = numpy.concatenate([effective, numpy.zeros(length - len(effective), dtype=bool)])
        return effective[:length]
