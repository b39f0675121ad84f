"""Host-side mirror of the hot-path entry `VoiceChanger.convert_from_acoustic_feature`
(/root/reference/realtime_voice_conversion/yukarin_wrapper/voice_changer.py:9-42): same constructor arguments,
same attribute names (`acoustic_converter`, `super_resolution`, `threshold`, `output_sampling_rate` -- read by
ConvertStream at realtime_voice_conversion/stream/convert_stream.py:15-27), same step order and in-place
`sp += 1e-16` / float32 cast, so that the reference's own class and this one are interchangeable.  Added on top:
`convert_windows`, which runs the two CNNs for several independent windows in one GPU batch (chunk parallelism
inside a GPU)."""
from typing import List, Optional

import numpy

SP_FLOOR = 1e-16            # voice_changer.py:39


class VoiceChanger(object):
    def __init__(self, acoustic_converter, super_resolution, threshold: float = 60, output_sampling_rate: Optional[int] = None) -> None:
        self.acoustic_converter = acoustic_converter
        self.super_resolution = super_resolution
        self.threshold = threshold
        self.output_sampling_rate = (super_resolution.config.dataset.param.voice_param.sample_rate
                                     if output_sampling_rate is None else output_sampling_rate)

    def _stage1(self, f_in):
        ac = self.acoustic_converter
        f_eff, effective = ac.separate_effective(wave=f_in.wave, feature=f_in, threshold=self.threshold)
        f_out = ac.convert(f_eff) if numpy.any(effective) else f_eff          # all-silent windows skip the CNN
        f_out = ac.combine_silent(effective=effective, feature=f_out)
        f_out = ac.decode_spectrogram(f_out)
        f_out.sp += SP_FLOOR
        return f_out

    def convert_from_acoustic_feature(self, f_in):
        f_out = self._stage1(f_in)
        f_out.sp = self.super_resolution.convert(f_out.sp.astype(numpy.float32))
        return f_out

    def convert_windows(self, f_ins: List) -> List:
        """Independent windows: stage-1 per window (its length is data dependent after the silence split),
        stage-2 for all equal-length windows in one batched GPU call."""
        outs = [self._stage1(f) for f in f_ins]
        lengths = {len(o.sp) for o in outs}
        sr = self.super_resolution
        if len(lengths) == 1 and hasattr(sr, '_get_net'):
            sp = numpy.stack([o.sp.astype(numpy.float32) for o in outs])
            res = sr._get_net(sp.shape[2]).convert(sp)
            for o, r in zip(outs, res):
                o.sp = r
        else:
            for o in outs:
                o.sp = sr.convert(o.sp.astype(numpy.float32))
        return outs
