"""`yukarin.f0_converter.F0Converter` ([MEM]; constructed at /root/reference/check.py:46-49 and
realtime_voice_conversion/converter/yukarin_converter.py:34-37).  Log-Gaussian normalised f0 transform
exp((sigma_t / sigma_i) (ln f0 - mu_i) + mu_t) on voiced frames; statistics are .npy files holding a pickled
dict with the mean / var of log f0."""
from pathlib import Path

import numpy

from .acoustic_feature import AcousticFeature


class F0Converter(object):
    def __init__(self, input_statistics: Path, target_statistics: Path) -> None:
        self.input_statistics_path = input_statistics
        self.target_statistics_path = target_statistics
        self.input_statistics = numpy.load(str(input_statistics), allow_pickle=True).item()
        self.target_statistics = numpy.load(str(target_statistics), allow_pickle=True).item()

    def convert(self, in_feature):
        f0_in = in_feature.f0 if isinstance(in_feature, AcousticFeature) else in_feature
        im, iv = self.input_statistics['mean'], self.input_statistics['var']
        tm, tv = self.target_statistics['mean'], self.target_statistics['var']
        f0 = numpy.copy(f0_in)
        nz = f0.nonzero()
        f0[nz] = numpy.exp((numpy.sqrt(tv) / numpy.sqrt(iv)) * (numpy.log(f0[nz]) - im) + tm)
        return AcousticFeature(f0=f0) if isinstance(in_feature, AcousticFeature) else f0
