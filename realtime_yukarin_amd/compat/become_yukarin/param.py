"""`become_yukarin.param.Param` ([MEM] defaults; the reference reads voice_param.sample_rate and
acoustic_feature_param.{frame_period, order}: /root/reference/realtime_voice_conversion/stream/convert_stream.py:16,25-27,
yukarin_wrapper/voice_changer.py:17, tests/test_convert_stream.py:5,32)."""
from typing import NamedTuple, Optional


class VoiceParam(NamedTuple):
    sample_rate: int = 24000
    top_db: Optional[float] = None
    pad_second: float = 0.0


class AcousticFeatureParam(NamedTuple):
    frame_period: int = 5
    order: int = 8
    alpha: float = 0.466
    f0_estimating_method: str = 'harvest'


class Param(NamedTuple):
    voice_param: VoiceParam = VoiceParam()
    acoustic_feature_param: AcousticFeatureParam = AcousticFeatureParam()
