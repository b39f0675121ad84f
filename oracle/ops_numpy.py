"""numpy restatement of the Chainer operators the convert hot path runs.  TEST INFRASTRUCTURE ONLY.

Every function restates a Chainer link/function that the (un-vendored) `yukarin` /
`become_yukarin` predictors call; the reference reaches them from
/root/reference/realtime_voice_conversion/yukarin_wrapper/voice_changer.py:33 (stage-1
`AcousticConverter.convert`) and :41 (stage-2 `SuperResolution.convert`).  Semantics follow
SURVEY.md §8(c) item 1:

* `ConvolutionND` / `Convolution2D`  = cross-correlation, W (Cout, Cin, k...), b (Cout,),
  out = floor((L + 2p - d(k-1) - 1)/s) + 1
* `DeconvolutionND` / `Deconvolution2D` = transposed conv, W (Cin, Cout, k...),
  out = s(L-1) + k - 2p
* `BatchNormalization` at inference: gamma (x - avg_mean) / sqrt(avg_var + eps) + beta, eps = 2e-5
* `F.leaky_relu` slope 0.2, `F.relu`, `F.sigmoid` (GLU gate), `F.dropout` = identity (train=False,
  /root/reference/realtime_voice_conversion/worker/convert_worker.py:31-32)

Arrays are NC(spatial...) like Chainer.  The arithmetic dtype is the dtype of `x`
(float32 = what the reference computes in; float64 = error-attribution variant).
"""
import numpy as np

BN_EPS = 2e-5
LRELU_SLOPE = 0.2


def _tup(v, n):
    if isinstance(v, (tuple, list)):
        assert len(v) == n
        return tuple(int(i) for i in v)
    return (int(v),) * n


def conv_nd(x, W, b=None, stride=1, pad=0, dilate=1):
    """Cross-correlation, N-d (n = W.ndim - 2).  x (B, Cin, *S), W (Cout, Cin, *k)."""
    n = W.ndim - 2
    assert x.ndim == n + 2 and x.shape[1] == W.shape[1]
    s, p, d = _tup(stride, n), _tup(pad, n), _tup(dilate, n)
    k = W.shape[2:]
    dt = x.dtype
    W = W.astype(dt)
    xp = np.pad(x, [(0, 0), (0, 0)] + [(pi, pi) for pi in p])
    out_sp = tuple((xp.shape[2 + i] - d[i] * (k[i] - 1) - 1) // s[i] + 1 for i in range(n))
    y = np.zeros((x.shape[0], W.shape[0]) + out_sp, dtype=dt)
    for tap in np.ndindex(*k):
        sl = tuple(slice(tap[i] * d[i], tap[i] * d[i] + s[i] * (out_sp[i] - 1) + 1, s[i]) for i in range(n))
        xs = xp[(slice(None), slice(None)) + sl]                       # (B, Cin, *out)
        w = W[(slice(None), slice(None)) + tap]                        # (Cout, Cin)
        y += np.tensordot(w, xs, axes=([1], [1])).swapaxes(0, 1).astype(dt)
    if b is not None:
        y += b.astype(dt).reshape((1, -1) + (1,) * n)
    return y


def deconv_nd(x, W, b=None, stride=1, pad=0):
    """Transposed convolution, N-d.  x (B, Cin, *S), W (Cin, Cout, *k); out = s(L-1)+k-2p."""
    n = W.ndim - 2
    assert x.ndim == n + 2 and x.shape[1] == W.shape[0]
    s, p = _tup(stride, n), _tup(pad, n)
    k = W.shape[2:]
    dt = x.dtype
    W = W.astype(dt)
    full_sp = tuple(s[i] * (x.shape[2 + i] - 1) + k[i] for i in range(n))
    full = np.zeros((x.shape[0], W.shape[1]) + full_sp, dtype=dt)
    for tap in np.ndindex(*k):
        sl = tuple(slice(tap[i], tap[i] + s[i] * (x.shape[2 + i] - 1) + 1, s[i]) for i in range(n))
        w = W[(slice(None), slice(None)) + tap]                        # (Cin, Cout)
        full[(slice(None), slice(None)) + sl] += np.tensordot(w, x, axes=([0], [1])).swapaxes(0, 1).astype(dt)
    crop = tuple(slice(p[i], full_sp[i] - p[i]) for i in range(n))
    y = np.ascontiguousarray(full[(slice(None), slice(None)) + crop])
    if b is not None:
        y += b.astype(dt).reshape((1, -1) + (1,) * n)
    return y


def batch_norm_inference(x, gamma, beta, avg_mean, avg_var, eps=BN_EPS):
    """Chainer `BatchNormalization.__call__` with `train=False` (fixed statistics)."""
    dt = x.dtype
    sh = (1, -1) + (1,) * (x.ndim - 2)
    inv = (1.0 / np.sqrt(avg_var.astype(dt) + dt.type(eps))).astype(dt)
    return (gamma.astype(dt).reshape(sh) * ((x - avg_mean.astype(dt).reshape(sh)) * inv.reshape(sh))
            + beta.astype(dt).reshape(sh)).astype(dt)


def leaky_relu(x, slope=LRELU_SLOPE):
    return np.where(x >= 0, x, x * x.dtype.type(slope)).astype(x.dtype)


def relu(x):
    return np.maximum(x, 0).astype(x.dtype)


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)


def glu(x):
    """Gated linear unit over the channel axis: first half * sigmoid(second half)."""
    c = x.shape[1] // 2
    return (x[:, :c] * sigmoid(x[:, c:])).astype(x.dtype)


def apply_act(x, act):
    if act in (None, 'none'):
        return x
    if act == 'lrelu':
        return leaky_relu(x)
    if act == 'relu':
        return relu(x)
    if act == 'glu':
        return glu(x)
    raise ValueError(act)
