"""realtime_yukarin_amd -- MI355X-native convert hot path of realtime-yukarin (stage-1 + stage-2 CNN forward).

Host code is Python (as in the reference); all arithmetic of the path runs in hand-written gfx950 HIP
kernels behind the C ABI of libry355.so (include/ry355.h).  No CPU fallback exists.
"""
from .netspec import NetDesc, pad_frames, param_count, param_list  # noqa: F401
