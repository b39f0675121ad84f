"""Stub of the only Chainer surface the reference touches directly:
`chainer.global_config.enable_backprop = False; chainer.global_config.train = False`
(/root/reference/realtime_voice_conversion/worker/convert_worker.py:6,31-32).  The MI355X path is
inference-only, so both flags are accepted and ignored."""
import contextlib


class _GlobalConfig(object):
    enable_backprop = False
    train = False
    dtype = 'float32'


global_config = _GlobalConfig()
config = global_config


@contextlib.contextmanager
def using_config(name, value, config=global_config):
    old = getattr(config, name, None)
    setattr(config, name, value)
    try:
        yield
    finally:
        setattr(config, name, old)
