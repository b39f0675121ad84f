"""torch-CPU (oneDNN, fp32) restatement of the same graph.  TEST INFRASTRUCTURE ONLY.

Second, independent implementation used (a) to pin `oracle.unet` (two implementations agreeing is
our substitute for the missing reference goldens) and (b) as the
`cpu_baseline` leg of bench.py ("CPU restatement (torch/oneDNN), not Chainer").
Weight layouts are identical to Chainer's: conv (Cout,Cin,k...), deconv (Cin,Cout,k...);
`eps=2e-5` is passed explicitly (torch's default differs).
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 2e-5
LRELU_SLOPE = 0.2


def _t(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


class TorchUNet:
    """Holds the K-list parameters as torch tensors; `forward` mirrors `oracle.unet.unet_forward`."""

    def __init__(self, P, extensive_layers=8, dtype=torch.float32, glu=False):
        self.e = int(extensive_layers)
        self.glu = bool(glu)                  # stage-1 glu_generator variant (UNVERIFIED [MEM], see oracle.unet.unet_forward)
        self.dtype = dtype
        self.P = {k: _t(v, dtype) for k, v in P.items() if not k.endswith('/N')}
        self.nd = self.P['encoder/c0/W'].dim() - 2

    def _conv(self, x, W, b, s, p):
        return (F.conv1d if self.nd == 1 else F.conv2d)(x, W, b, stride=s, padding=p)

    def _deconv(self, x, W, b, s, p):
        return (F.conv_transpose1d if self.nd == 1 else F.conv_transpose2d)(x, W, b, stride=s, padding=p)

    def _cbr(self, x, prefix, sample, act):
        P = self.P
        W, b = P[prefix + '/c/W'], P[prefix + '/c/b']
        if sample == 'down':
            h = self._conv(x, W, b, 2, 1)
        elif sample == 'up':
            h = self._deconv(x, W, b, 2, 1)
        else:
            h = self._conv(x, W, b, 1, 0)
        bn = prefix + '/batchnorm/'
        h = F.batch_norm(h, P[bn + 'avg_mean'], P[bn + 'avg_var'], P[bn + 'gamma'], P[bn + 'beta'],
                         training=False, eps=BN_EPS)
        if self.glu:
            return F.glu(h, dim=1)            # h[:, :C] * sigmoid(h[:, C:])
        return F.leaky_relu(h, LRELU_SLOPE) if act == 'lrelu' else F.relu(h)

    @torch.no_grad()
    def forward(self, x):
        e, P = self.e, self.P
        ep = 1 if e > 0 else 0
        hs = [F.leaky_relu(self._conv(x, P['encoder/c0/W'], P['encoder/c0/b'], 1, ep), LRELU_SLOPE)]
        for i in range(1, 8):
            hs.append(self._cbr(hs[i - 1], 'encoder/c%d' % i, 'down' if i < e else 'same', 'lrelu'))
        h = self._cbr(hs[7], 'decoder/c0', 'up' if 7 < e else 'same', 'relu')
        for j in range(1, 8):
            h = torch.cat([h, hs[7 - j]], dim=1)
            if j < 7:
                h = self._cbr(h, 'decoder/c%d' % j, 'up' if (7 - j) < e else 'same', 'relu')
            else:
                h = self._conv(h, P['decoder/c7/W'], P['decoder/c7/b'], 1, ep)
        return h

    def forward_np(self, x):
        return self.forward(_t(x, self.dtype)).numpy()


def stage1_convert_core(net, x_nc):
    n = x_nc.shape[0]
    pad = 128 - n % 128
    x = np.pad(x_nc.T, [(0, 0), (0, pad)], mode='minimum')
    y = net.forward_np(x[np.newaxis])[0]
    return np.ascontiguousarray(y[:, :-pad].T)


def stage2_convert(net, sp):
    n = sp.shape[0]
    pad = 128 - n % 128
    x = np.pad(sp, [(0, pad), (0, 0)], mode='minimum')
    x = np.log(x)[:, :-1]
    y = net.forward_np(x[np.newaxis, np.newaxis])[0, 0]
    y = np.pad(y, [(0, 0), (0, 1)], mode='edge')
    return np.ascontiguousarray(np.exp(y)[:-pad])
