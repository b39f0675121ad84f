"""`become_yukarin` import surface used by realtime-yukarin, backed by libry355.so."""
from .super_resolution import SuperResolution  # noqa: F401
