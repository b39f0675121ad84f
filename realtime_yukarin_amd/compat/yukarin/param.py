"""`yukarin.param.AcousticParam` ([MEM] field list and defaults; the reference reads sampling_rate,
frame_period, order, alpha, f0_floor, f0_ceil, fft_length, dtype: /root/reference/realtime_voice_conversion/
yukarin_wrapper/vocoder.py:30-36, stream/encode_stream.py:18-24; frame_period default 5 is pinned by
/root/reference/tests/test_convert_stream.py:46-47)."""
from typing import NamedTuple, Optional


class AcousticParam(NamedTuple):
    sampling_rate: int = 24000
    pad_second: float = 0
    threshold_db: Optional[float] = None
    frame_period: int = 5
    order: int = 8
    alpha: float = 0.466
    f0_floor: float = 71
    f0_ceil: float = 800
    fft_length: int = 1024
    dtype: str = 'float32'
