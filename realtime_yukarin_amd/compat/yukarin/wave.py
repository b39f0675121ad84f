"""`yukarin.wave.Wave` ([MEM]; used at /root/reference/realtime_voice_conversion/yukarin_wrapper/
acoustic_feature_wrapper.py:23,45,52 and stream/encode_stream.py:35)."""
import numpy


class Wave(object):
    def __init__(self, wave: numpy.ndarray, sampling_rate: int) -> None:
        self.wave = wave
        self.sampling_rate = sampling_rate

    def get_hop_and_length(self, frame_period: float):
        hop = self.sampling_rate * frame_period // 1000
        length = int(len(self.wave) / hop) + 1
        return int(hop), length

    def get_effective_frame(self, threshold_db: float, fft_length: int, frame_period: float) -> numpy.ndarray:
        """Per-frame mask: frame power (dB relative to the loudest frame) above -threshold_db.
        Restates librosa.feature.rms(center=True, pad reflect) ** 2 -> power_to_db(ref=max, top_db=None) [MEM]."""
        hop, length = self.get_hop_and_length(frame_period)
        y = numpy.asarray(self.wave, dtype=numpy.float64)
        if len(y) == 0:
            return numpy.zeros(0, dtype=bool)
        half = fft_length // 2
        mode = 'reflect' if len(y) > half else 'edge'
        yp = numpy.pad(y, half, mode=mode)
        n_frames = 1 + (len(yp) - fft_length) // hop
        csum = numpy.concatenate([[0.0], numpy.cumsum(yp * yp)])
        starts = numpy.arange(n_frames) * hop
        mse = (csum[starts + fft_length] - csum[starts]) / fft_length
        ref = mse.max()
        amin = 1e-10
        db = 10.0 * numpy.log10(numpy.maximum(amin, mse)) - 10.0 * numpy.log10(numpy.maximum(amin, ref))
        effective = db > -threshold_db
        if len(effective) < length:
            effective = numpy.concatenate([effective, numpy.zeros(length - len(effective), dtype=bool)])
        return effective[:length]
