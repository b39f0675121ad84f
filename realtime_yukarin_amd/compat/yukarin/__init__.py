"""`yukarin` import surface used by realtime-yukarin, backed by libry355.so."""
from .acoustic_feature import AcousticFeature  # noqa: F401
from .wave import Wave  # noqa: F401
from .acoustic_converter import AcousticConverter  # noqa: F401
