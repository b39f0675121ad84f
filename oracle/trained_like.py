"""Weights with the statistics of a TRAINED predictor, for parity tests.  TEST INFRASTRUCTURE ONLY.

The canonical synthetic weights (pix2pix init, W ~ N(0, 0.02), b = 0, gamma ~ N(1, 0.02), beta = 0,
avg_var ~ U(0.5, 1.5)) are the kindest possible input for a folded-BatchNormalization epilogue: every per-channel scale is ~ 1.
A trained Chainer model ([MEM]: `chainer.links.BatchNormalization` keeps running `avg_mean` / `avg_var` of what the convolution in
front of it actually produced; loaded by the constructors called at /root/reference/realtime_voice_conversion/converter/
yukarin_converter.py:40-55) looks different: the convolution outputs of different channels differ by orders of magnitude and carry
offsets, `avg_var` therefore spans decades, |gamma| is not near 1, biases and beta are non-zero -- and BN brings every channel back to
unit scale, which is what keeps the 16-layer chain from exploding.

`calibrated_params` builds exactly that: per layer, filters N(0, 0.02) x a per-output-channel gain drawn log-uniformly from
10^-1.5 .. 10^1 (so avg_var spans ~ 5 decades), bias N(0, 0.1) x gain; then the layer runs on a calibration input (torch, CPU) and
`avg_mean` / `avg_var` are set to the measured per-channel statistics of its convolution output, perturbed by a few percent as running
averages are; gamma ~ N(1, 0.5) (sign changes included), beta ~ N(0, 0.1).  The result is an ordinary K-list dict that every
implementation (numpy / torch / C oracle, the emulator, the HIP path) loads like any other weight set."""
import numpy
import torch
import torch.nn.functional as F

from .unet import DEC_IN, DEC_OUT, ENC_CH

LRELU, EPS = 0.2, 2e-5


def enc_sample(desc, i):
    return 'down' if i < desc.extensive_layers else 'same'


def dec_sample(desc, j):
    return 'up' if (7 - j) < desc.extensive_layers else 'same'


def calibrated_params(desc, seed: int, x_calib: numpy.ndarray):
    """desc: anything with ndim / in_ch / out_ch / base / extensive_layers (e.g. netspec.NetDesc).  x_calib: (B, in_ch, T) for stage 1, (B, 1, T, W) for stage 2 (T a multiple of 128) -- what the predictor sees after the
    wrapper's pad / log.  Returns the K-list dict (float32)."""
    rng = numpy.random.default_rng(seed)
    nd, B = desc.ndim, desc.base
    conv = F.conv1d if nd == 1 else F.conv2d
    deconv = F.conv_transpose1d if nd == 1 else F.conv_transpose2d
    end_k = 3 if desc.extensive_layers > 0 else 1
    P = {}

    def t(a):
        return torch.from_numpy(numpy.ascontiguousarray(a, dtype=numpy.float32))

    def draw_w(shape, out_axis):
        co = shape[out_axis]
        gain = 10.0 ** rng.uniform(-1.5, 1.0, co)
        gshape = [1] * len(shape); gshape[out_axis] = co
        w = rng.normal(0.0, 0.02, shape) * gain.reshape(gshape)
        b = rng.normal(0.0, 0.1, co) * gain
        return w.astype(numpy.float32), b.astype(numpy.float32)

    def cbr(x, prefix, sample, ci, co, act):
        if sample == 'up':
            W, b = draw_w((ci, co) + (4,) * nd, 1)
            h = deconv(x, t(W), t(b), stride=2, padding=1)
        elif sample == 'down':
            W, b = draw_w((co, ci) + (4,) * nd, 0)
            h = conv(x, t(W), t(b), stride=2, padding=1)
        else:
            W, b = draw_w((co, ci) + (1,) * nd, 0)
            h = conv(x, t(W), t(b))
        axes = [0] + list(range(2, h.dim()))
        mean = h.mean(dim=axes).numpy().astype(numpy.float64)
        var = h.var(dim=axes, unbiased=False).numpy().astype(numpy.float64)
        n_per_ch = h.numel() // h.shape[1]
        if n_per_ch < 16:                                       # the bottom of the U-Net: too few samples for a variance
            var = var + (0.1 * numpy.abs(mean) + 1e-3) ** 2
        mean = mean + rng.normal(0.0, 0.05, co) * numpy.sqrt(var)
        var = var * rng.uniform(0.8, 1.25, co)
        gamma = rng.normal(1.0, 0.5, co)
        beta = rng.normal(0.0, 0.1, co)
        P[prefix + '/c/W'], P[prefix + '/c/b'] = W, b
        for k, v in (('gamma', gamma), ('beta', beta), ('avg_mean', mean), ('avg_var', var)):
            P[prefix + '/batchnorm/' + k] = v.astype(numpy.float32)
        h = F.batch_norm(h, t(mean), t(var), t(gamma), t(beta), training=False, eps=EPS)
        return F.leaky_relu(h, LRELU) if act == 'lrelu' else F.relu(h)

    with torch.no_grad():
        x = t(x_calib)
        W0 = rng.normal(0.0, 0.02, (B, desc.in_ch) + (end_k,) * nd).astype(numpy.float32)
        b0 = rng.normal(0.0, 0.1, B).astype(numpy.float32)
        P['encoder/c0/W'], P['encoder/c0/b'] = W0, b0
        hs = [F.leaky_relu(conv(x, t(W0), t(b0), padding=end_k // 2), LRELU)]
        for i in range(1, 8):
            hs.append(cbr(hs[i - 1], 'encoder/c%d' % i, enc_sample(desc, i), ENC_CH[i - 1] * B, ENC_CH[i] * B, 'lrelu'))
        h = cbr(hs[7], 'decoder/c0', dec_sample(desc, 0), DEC_IN[0] * B, DEC_OUT[0] * B, 'relu')
        for j in range(1, 7):
            h = torch.cat([h, hs[7 - j]], dim=1)
            h = cbr(h, 'decoder/c%d' % j, dec_sample(desc, j), DEC_IN[j] * B, DEC_OUT[j] * B, 'relu')
        P['decoder/c7/W'] = rng.normal(0.0, 0.02, (desc.out_ch, 2 * B) + (end_k,) * nd).astype(numpy.float32)
        P['decoder/c7/b'] = rng.normal(0.0, 0.1, desc.out_ch).astype(numpy.float32)
    return P


def describe(P):
    """(min, max) of avg_var, of |gamma / sqrt(avg_var + eps)| (the folded scale) and of |gamma| over all BN layers."""
    var = numpy.concatenate([v for k, v in P.items() if k.endswith('avg_var')])
    gam = numpy.concatenate([v for k, v in P.items() if k.endswith('gamma')])
    sc = numpy.abs(gam) / numpy.sqrt(var + EPS)
    return dict(avg_var=(float(var.min()), float(var.max())), folded_scale=(float(sc.min()), float(sc.max())),
                abs_gamma=(float(numpy.abs(gam).min()), float(numpy.abs(gam).max())))
