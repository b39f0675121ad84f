"""ctypes face of oracle/ops_ref.c with the call signatures of ops_numpy.  TEST INFRASTRUCTURE ONLY.

`build()` compiles ops_ref.c with `gcc -O2 -fopenmp` into oracle/_build/libops_ref.so (git-ignored, travels to the GPU box
with the snapshot; rebuilt there if missing -- gcc is in the image).  `unet.unet_forward(..., ops=c_ref)` runs the two
predictors on it."""
import ctypes
import shutil
import subprocess
from pathlib import Path

import numpy as np

from . import ops_numpy

HERE = Path(__file__).resolve().parent
SRC = HERE / 'ops_ref.c'
LIB = HERE / '_build' / 'libops_ref.so'
BN_EPS, LRELU_SLOPE = ops_numpy.BN_EPS, ops_numpy.LRELU_SLOPE
ACC64 = False            # True: sums in double (error-attribution variant)
_dll = None
_FP = ctypes.POINTER(ctypes.c_float)


def build(force: bool = False) -> Path:
    if force or not LIB.exists() or LIB.stat().st_mtime < SRC.stat().st_mtime:
        cc = shutil.which('gcc') or shutil.which('cc')
        if cc is None:
            raise RuntimeError('gcc not found: cannot build the C oracle')
        LIB.parent.mkdir(exist_ok=True)
        subprocess.run([cc, '-O2', '-fopenmp', '-shared', '-fPIC', '-std=c99', str(SRC), '-o', str(LIB), '-lm'], check=True)
    return LIB


def _lib():
    global _dll
    if _dll is None:
        _dll = ctypes.CDLL(str(build()))
        I, F = ctypes.c_int, ctypes.c_float
        _dll.ry_ref_conv.argtypes = [_FP, I, I, I, I, _FP, _FP, I, I, I, I, I, I, I, I, I, I, _FP]
        _dll.ry_ref_deconv.argtypes = [_FP, I, I, I, I, _FP, _FP, I, I, I, I, I, I, I, I, _FP]
        _dll.ry_ref_bn_act.argtypes = [_FP, I, I, ctypes.c_size_t, _FP, _FP, _FP, _FP, F, I, F]
        for f in (_dll.ry_ref_conv, _dll.ry_ref_deconv, _dll.ry_ref_bn_act):
            f.restype = None
    return _dll


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(_FP) if a is not None else ctypes.cast(None, _FP)


def _hw(v, n):
    """(h, w) form of a per-axis argument; 1-D layers run as H = 1 with neutral height parameters."""
    t = ops_numpy._tup(v, n)
    return (None, t[0]) if n == 1 else t


def conv_nd(x, W, b=None, stride=1, pad=0, dilate=1):
    n = W.ndim - 2
    assert n in (1, 2) and x.ndim == n + 2 and x.shape[1] == W.shape[1]
    x, W = _f(x), _f(W)
    b = _f(b) if b is not None else None
    (sh, sw), (ph, pw), (dh, dw) = _hw(stride, n), _hw(pad, n), _hw(dilate, n)
    if n == 1:
        sh, ph, dh = 1, 0, 1
    B, Cin = x.shape[:2]
    H, Wd = (1, x.shape[2]) if n == 1 else x.shape[2:]
    kh, kw = (1, W.shape[2]) if n == 1 else W.shape[2:]
    Ho, Wo = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1, (Wd + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    y = np.empty((B, W.shape[0], Ho, Wo), np.float32)
    _lib().ry_ref_conv(_p(x), B, Cin, H, Wd, _p(W), _p(b), W.shape[0], kh, kw, sh, sw, ph, pw, dh, dw, int(ACC64), _p(y))
    return y[:, :, 0] if n == 1 else y


def deconv_nd(x, W, b=None, stride=1, pad=0):
    n = W.ndim - 2
    assert n in (1, 2) and x.ndim == n + 2 and x.shape[1] == W.shape[0]
    x, W = _f(x), _f(W)
    b = _f(b) if b is not None else None
    (sh, sw), (ph, pw) = _hw(stride, n), _hw(pad, n)
    if n == 1:
        sh, ph = 1, 0
    B, Cin = x.shape[:2]
    H, Wd = (1, x.shape[2]) if n == 1 else x.shape[2:]
    kh, kw = (1, W.shape[2]) if n == 1 else W.shape[2:]
    Ho, Wo = sh * (H - 1) + kh - 2 * ph, sw * (Wd - 1) + kw - 2 * pw
    y = np.empty((B, W.shape[1], Ho, Wo), np.float32)
    _lib().ry_ref_deconv(_p(x), B, Cin, H, Wd, _p(W), _p(b), W.shape[1], kh, kw, sh, sw, ph, pw, int(ACC64), _p(y))
    return y[:, :, 0] if n == 1 else y


def _bn_act(x, bn, act):
    y = _f(x).copy()
    S = int(np.prod(y.shape[2:]))
    g, be, m, v = [(_f(a) if a is not None else None) for a in (bn or (None,) * 4)]
    code = {None: 0, 'none': 0, 'lrelu': 1, 'relu': 2}[act]
    _lib().ry_ref_bn_act(_p(y), y.shape[0], y.shape[1], S, _p(g), _p(be), _p(m), _p(v), BN_EPS, code, LRELU_SLOPE)
    return y


def batch_norm_inference(x, gamma, beta, avg_mean, avg_var, eps=BN_EPS):
    assert eps == BN_EPS
    return _bn_act(x, (gamma, beta, avg_mean, avg_var), None)


def leaky_relu(x, slope=LRELU_SLOPE):
    assert slope == LRELU_SLOPE
    return _bn_act(x, None, 'lrelu')


def relu(x):
    return _bn_act(x, None, 'relu')


def apply_act(x, act):
    if act == 'glu':
        return ops_numpy.glu(_f(x))
    return _bn_act(x, None, act)
