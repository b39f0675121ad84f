"""Predictor weights: Chainer `save_npz` reader, strict K-list validation, flat blob, synthetic init.

The reference loads weights with `chainer.serializers.load_npz(model_path, model)` inside the
dependency constructors ([MEM]; call sites /root/reference/realtime_voice_conversion/converter/
yukarin_converter.py:40-55, /root/reference/check.py:54-63).  Trained models are not shipped
(/root/reference/README.md:22-40), so `synthetic_params` provides the seeded canonical weights
used by tests and bench (conv W ~ N(0, 0.02), b = 0, BN gamma ~ N(1, 0.02),
beta = 0, avg_mean ~ N(0, 0.1), avg_var ~ U(0.5, 1.5)).
"""
from pathlib import Path
from typing import Dict

import numpy

from .netspec import NetDesc, param_list


def synthetic_params(desc: NetDesc, seed: int, bias_std: float = 0.0) -> Dict[str, numpy.ndarray]:
    """Seeded random-init weights with the architecture's shapes (no checkpoint is available)."""
    rng = numpy.random.default_rng(seed)
    P = {}
    for key, shape in param_list(desc):
        leaf = key.rsplit('/', 1)[1]
        if leaf == 'W':
            a = rng.normal(0.0, 0.02, size=shape)
        elif leaf == 'b':
            a = rng.normal(0.0, bias_std, size=shape) if bias_std > 0 else numpy.zeros(shape)
        elif leaf == 'gamma':
            a = rng.normal(1.0, 0.02, size=shape)
        elif leaf == 'beta':
            a = rng.normal(0.0, bias_std, size=shape) if bias_std > 0 else numpy.zeros(shape)
        elif leaf == 'avg_mean':
            a = rng.normal(0.0, 0.1, size=shape)
        elif leaf == 'avg_var':
            a = rng.uniform(0.5, 1.5, size=shape)
        else:  # pragma: no cover
            raise KeyError(key)
        P[key] = a.astype(numpy.float32)
    return P


def validate_params(desc: NetDesc, P: Dict[str, numpy.ndarray]) -> None:
    """Refuse anything that is not exactly the predictor's K-list (never silently mis-map)."""
    want = dict(param_list(desc))
    have = {k: v for k, v in P.items() if not k.endswith('/N')}      # BN sample counter is unused
    missing = sorted(set(want) - set(have))
    extra = sorted(set(have) - set(want))
    if missing or extra:
        raise ValueError('weight keys do not match the predictor: missing=%s unexpected=%s' % (missing, extra))
    for k, shape in want.items():
        if tuple(have[k].shape) != tuple(shape):
            raise ValueError('weight %s has shape %s, config needs %s' % (k, tuple(have[k].shape), shape))


def load_npz(desc: NetDesc, path: Path) -> Dict[str, numpy.ndarray]:
    """Read a Chainer `save_npz` file of the bare predictor (keys = K-list)."""
    with numpy.load(str(path)) as z:
        P = {k: numpy.asarray(z[k]) for k in z.files}
    validate_params(desc, P)
    return {k: P[k].astype(numpy.float32) for k, _ in param_list(desc)}


def save_npz(path: Path, P: Dict[str, numpy.ndarray]) -> None:
    numpy.savez(str(path), **P)


def flatten_params(desc: NetDesc, P: Dict[str, numpy.ndarray]) -> numpy.ndarray:
    """Canonical flat float32 blob (K-list order) = the `weights_flat` argument of `ry_net_create`."""
    validate_params(desc, P)
    return numpy.concatenate([numpy.ascontiguousarray(P[k], dtype=numpy.float32).ravel()
                              for k, _ in param_list(desc)])


def unflatten_params(desc: NetDesc, blob: numpy.ndarray) -> Dict[str, numpy.ndarray]:
    P, off = {}, 0
    for k, shape in param_list(desc):
        n = int(numpy.prod(shape))
        P[k] = numpy.asarray(blob[off:off + n], dtype=numpy.float32).reshape(shape)
        off += n
    if off != blob.size:
        raise ValueError('blob has %d floats, predictor needs %d' % (blob.size, off))
    return P
