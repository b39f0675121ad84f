"""`yukarin.acoustic_feature.AcousticFeature` container ([MEM] restatement of the methods the reference calls:
get_sizes / silent / concatenate / pick / astype_only_float / extract: /root/reference/realtime_voice_conversion/
yukarin_wrapper/acoustic_feature_wrapper.py:9-62, segment/feature_segment.py:26-37, yukarin_wrapper/vocoder.py:28-58).
Plain numpy holders, (N, dim) row-major per key.  `extract` (WORLD analysis) is outside the accelerated path
 and needs the optional pyworld/pysptk packages."""
from typing import Dict, Iterable, List

import numpy

_NAN = numpy.nan


def cheaptrick_fft_size(sampling_rate: int, f0_floor: float = 71.0) -> int:
    """WORLD GetFFTSizeForCheapTrick: 2 ** (1 + floor(log2(3 fs / f0_floor + 1)))  (= 1024 at 16/24 kHz)."""
    return int(2 ** (1 + int(numpy.log2(3.0 * sampling_rate / f0_floor + 1))))


def num_aperiodicities(sampling_rate: int) -> int:
    """WORLD GetNumberOfAperiodicities: floor(min(15000, fs/2 - 3000) / 3000)."""
    return int(min(15000.0, sampling_rate / 2.0 - 3000.0) / 3000.0)


class AcousticFeature(object):
    all_keys = ('f0', 'sp', 'ap', 'coded_ap', 'mc', 'voiced')

    def __init__(self, f0=_NAN, sp=_NAN, ap=_NAN, coded_ap=_NAN, mc=_NAN, voiced=_NAN) -> None:
        self.f0 = f0
        self.sp = sp
        self.ap = ap
        self.coded_ap = coded_ap
        self.mc = mc
        self.voiced = voiced

    # legacy names read by the non-realtime Vocoder.decode (vocoder.py:57-58)
    @property
    def spectrogram(self):
        return self.sp

    @property
    def aperiodicity(self):
        return self.ap

    @staticmethod
    def _is_array(v) -> bool:
        return isinstance(v, numpy.ndarray)

    def astype(self, dtype):
        return AcousticFeature(**{k: (v.astype(dtype) if self._is_array(v) else v) for k, v in self.__dict__.items()
                                  if k in self.all_keys})

    def astype_only_float(self, dtype):
        out = {}
        for k in self.all_keys:
            v = getattr(self, k)
            if self._is_array(v) and k != 'voiced' and v.dtype.kind == 'f':
                v = v.astype(dtype)
            out[k] = v
        return AcousticFeature(**out)

    def validate(self):
        lengths = [len(getattr(self, k)) for k in self.all_keys if self._is_array(getattr(self, k))]
        assert len(set(lengths)) <= 1, 'features have different lengths: %s' % lengths

    @staticmethod
    def get_sizes(sampling_rate: int, order: int) -> Dict[str, int]:
        fft = cheaptrick_fft_size(sampling_rate)
        return dict(f0=1, sp=fft // 2 + 1, ap=fft // 2 + 1, mc=order + 1, voiced=1, coded_ap=num_aperiodicities(sampling_rate))

    @staticmethod
    def silent(length: int, sizes: Dict[str, int], keys: Iterable[str]):
        d = {}
        for k in keys:
            d[k] = numpy.zeros((length, sizes[k]), dtype=bool if k == 'voiced' else numpy.float32)
        return AcousticFeature(**d)

    @staticmethod
    def concatenate(fs: List['AcousticFeature'], keys: Iterable[str]):
        # members left at their NaN default (feature not extracted) stay NaN, as the reference's tests rely on
        return AcousticFeature(**{k: (numpy.concatenate([getattr(f, k) for f in fs]) if AcousticFeature._is_array(getattr(fs[0], k)) else _NAN)
                                  for k in keys})

    def pick(self, first: int, last: int, keys: Iterable[str]):
        return AcousticFeature(**{k: (getattr(self, k)[first:last] if self._is_array(getattr(self, k)) else _NAN) for k in keys})

    def indexing(self, index: numpy.ndarray):
        return AcousticFeature(**{k: (getattr(self, k)[index] if self._is_array(getattr(self, k)) else getattr(self, k))
                                  for k in self.all_keys})

    def indexing_set(self, index: numpy.ndarray, feature: 'AcousticFeature'):
        for k in self.all_keys:
            dst, src = getattr(self, k), getattr(feature, k)
            if self._is_array(dst) and self._is_array(src):
                dst[index] = src

    @classmethod
    def extract_f0(cls, x: numpy.ndarray, fs: int, frame_period: int, f0_floor: float, f0_ceil: float):
        import pyworld  # optional, CPU WORLD; not part of the accelerated path
        f0, t = pyworld.harvest(x, fs, frame_period=frame_period, f0_floor=f0_floor, f0_ceil=f0_ceil)
        return f0, t

    @classmethod
    def extract(cls, wave, frame_period, f0_floor, f0_ceil, fft_length, order, alpha, dtype):
        """WORLD analysis (f0 -> CheapTrick sp -> D4C ap -> sp2mc).  OUT OF SCOPE of the MI355X path
        (SURVEY.md section 3.4); delegates to pyworld/pysptk when they are installed."""
        try:
            import pyworld
            import pysptk
        except ImportError as e:  # pragma: no cover
            raise NotImplementedError('AcousticFeature.extract needs pyworld and pysptk (WORLD analysis is a CPU stage '
                                      'outside the accelerated convert path)') from e
        x = wave.wave.astype(numpy.float64)
        fs = wave.sampling_rate
        f0, t = cls.extract_f0(x=x, fs=fs, frame_period=frame_period, f0_floor=f0_floor, f0_ceil=f0_ceil)
        sp = pyworld.cheaptrick(x, f0, t, fs, fft_size=fft_length)
        ap = pyworld.d4c(x, f0, t, fs, fft_size=fft_length)
        mc = pysptk.sp2mc(sp, order=order, alpha=alpha)
        coded_ap = pyworld.code_aperiodicity(ap, fs)
        voiced = ~(f0 == 0)
        feature = AcousticFeature(f0=f0[:, None], sp=sp, ap=ap, coded_ap=coded_ap, mc=mc, voiced=voiced[:, None])
        feature = feature.astype_only_float(dtype)
        feature.validate()
        return feature
