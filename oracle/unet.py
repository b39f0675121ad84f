"""numpy restatement of the two predictors and their convert() wrappers.  TEST INFRASTRUCTURE ONLY.

Topology follows SURVEY.md §8(a) rows A3/A7 and §8(c) items 2-5 ([MEM] restatement of
`yukarin.model.Predictor` and `become_yukarin.model.sr_model.SRPredictor`; sources are not under
/root/reference, call sites are /root/reference/realtime_voice_conversion/yukarin_wrapper/
voice_changer.py:33,41 and converter/yukarin_converter.py:40-55).  Parameters are a dict keyed by
the Chainer `save_npz` names (K-list):

    encoder/c0/{W,b}
    encoder/c{1..7}/c/{W,b}   encoder/c{1..7}/batchnorm/{gamma,beta,avg_mean,avg_var,N}
    decoder/c{0..6}/c/{W,b}   decoder/c{0..6}/batchnorm/{...}
    decoder/c7/{W,b}

Layer rule (extensive_layers = e): encoder c_i (i=1..7) is a k4 s2 p1 "down" conv iff i < e, else a
k1 "same" conv; decoder c_j (j=0..6) mirrors encoder c_{7-j} ("up" k4 s2 p1 deconv iff 7-j < e);
c0 / decoder c7 are k3 p1 if e > 0 else k1.
"""
import numpy as np

from . import ops_numpy as ops

ENC_CH = [1, 2, 4, 8, 8, 8, 8, 8]            # encoder c_i output channels / base
DEC_IN = [8, 16, 16, 16, 16, 8, 4]           # decoder c_j input channels / base (after concat)
DEC_OUT = [8, 8, 8, 8, 4, 2, 1]              # decoder c_j output channels / base


def _cbr(x, P, prefix, sample, act, ops=ops):
    """`CBR.__call__`: conv/deconv -> BatchNormalization -> (dropout = identity) -> activation."""
    W, b = P[prefix + '/c/W'], P[prefix + '/c/b']
    if sample == 'down':
        h = ops.conv_nd(x, W, b, stride=2, pad=1)
    elif sample == 'up':
        h = ops.deconv_nd(x, W, b, stride=2, pad=1)
    else:
        h = ops.conv_nd(x, W, b, stride=1, pad=0)
    bn = prefix + '/batchnorm/'
    h = ops.batch_norm_inference(h, P[bn + 'gamma'], P[bn + 'beta'], P[bn + 'avg_mean'], P[bn + 'avg_var'])
    return ops.apply_act(h, act)


def unet_forward(x, P, extensive_layers=8, return_all=False, ops=ops, glu=False):
    """`Predictor.__call__` / `SRPredictor.__call__`: x (B, in_ch, *spatial) -> (B, out_ch, *spatial).

    The spatial rank is taken from the weights (1 -> stage-1, 2 -> stage-2).  `ops` selects the operator restatement:
    `ops_numpy` (default) or `c_ref` (the plain-C loop nests of ops_ref.c).

    glu=True: the stage-1 `model.glu_generator` variant as THIS repository reads it -- UNVERIFIED [MEM], the upstream class is in the
    un-vendored `yukarin` package: every conv + BatchNormalization block computes twice the channels and is gated, h[:C] * sigmoid(h[C:])
    (`chainer.functions.glu` along the channel axis), in place of leaky_relu (encoder) / relu (decoder); c0 and the last layer as before."""
    e = int(extensive_layers)
    end_pad = 1 if e > 0 else 0
    act_e, act_d = ('glu', 'glu') if glu else ('lrelu', 'relu')
    hs = [ops.leaky_relu(ops.conv_nd(x, P['encoder/c0/W'], P['encoder/c0/b'], stride=1, pad=end_pad))]
    for i in range(1, 8):
        hs.append(_cbr(hs[i - 1], P, 'encoder/c%d' % i, 'down' if i < e else 'same', act_e, ops))
    h = _cbr(hs[7], P, 'decoder/c0', 'up' if 7 < e else 'same', act_d, ops)
    acts = {'enc': hs, 'dec': [h]}
    for j in range(1, 8):
        h = np.concatenate([h, hs[7 - j]], axis=1)
        if j < 7:
            h = _cbr(h, P, 'decoder/c%d' % j, 'up' if (7 - j) < e else 'same', act_d, ops)
            acts['dec'].append(h)
        else:
            h = ops.conv_nd(h, P['decoder/c7/W'], P['decoder/c7/b'], stride=1, pad=end_pad)
    if return_all:
        return h, acts
    return h


def pad_frames(n):
    """Both wrappers: pad = 128 - n % 128 (a full extra 128 when n % 128 == 0)."""
    return 128 - n % 128


def stage1_convert_core(x_nc, P, extensive_layers=8, ops=ops, glu=False):
    """Array part of `AcousticConverter.convert` ([MEM]):
    x_nc (N, C_in) = encode_feature(...) before the transpose -> (N, C_out).

    transpose -> numpy.pad(mode='minimum') along time -> batch axis -> Predictor -> crop -> transpose."""
    n = x_nc.shape[0]
    pad = pad_frames(n)
    x = np.pad(x_nc.T, [(0, 0), (0, pad)], mode='minimum')
    y = unet_forward(x[np.newaxis], P, extensive_layers, ops=ops, glu=glu)[0]
    return np.ascontiguousarray(y[:, :-pad].T)


def stage2_convert(sp, P, extensive_layers=8, ops=ops):
    """`SuperResolution.convert` ([MEM]): sp (N, F) float32 -> (N, F).

    pad 'minimum' along time -> log -> drop last bin -> (1,1,T,F-1) -> SRPredictor -> [0][0]
    -> pad 'edge' one bin -> exp -> crop."""
    n = sp.shape[0]
    pad = pad_frames(n)
    x = np.pad(sp, [(0, pad), (0, 0)], mode='minimum')
    x = np.log(x)[:, :-1]
    y = unet_forward(x[np.newaxis, np.newaxis], P, extensive_layers, ops=ops)[0, 0]
    y = np.pad(y, [(0, 0), (0, 1)], mode='edge')
    y = np.exp(y)
    return np.ascontiguousarray(y[:-pad])
