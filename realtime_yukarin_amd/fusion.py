"""Keeps the data on the GPU between the two CNNs when the UNCHANGED reference class drives the shims step by step.

The reference's `VoiceChanger.convert_from_acoustic_feature` (/root/reference/realtime_voice_conversion/yukarin_wrapper/
voice_changer.py:33-41) calls

    f_out = acoustic_converter.convert(f_in_effective)            # stage-1
    f_out = acoustic_converter.combine_silent(effective, f_out)
    f_out = acoustic_converter.decode_spectrogram(f_out)          # pysptk.mc2sp on the host
    f_out.sp += 1e-16
    f_out.sp = super_resolution.convert(f_out.sp.astype(numpy.float32))   # stage-2

Taken literally that is two H2D / D2H round trips and a host mc2sp per window.  The shims instead leave the converted rows on the
device after `convert` (`ry_vc_stage1`), carry a token through `combine_silent`, and let `decode_spectrogram` return a
`LazySpectrogram`: an object that remembers the floor added by `+=`, answers `astype`, and -- handed to the SuperResolution shim it was
built for -- triggers `ry_vc_stage2_from_mc`: combine_silent + mc2sp + floor + stage-2 on the device, one D2H of the result.  Anything
else that touches it (numpy functions, indexing, another SuperResolution) gets the real array (`ry_vc_mid_sp`, or the host formula if
the device rows have been overwritten since), so the object is indistinguishable from the array except for where the bytes live.
One small H2D (effective frames) + one small D2H (converted mc rows) + one D2H (spectrogram) per window; the reference is not edited.
`RY_FUSE_STEPS=0` switches it off (plain arrays at every step)."""
import os
import weakref

import numpy

from . import engine, sptk

_sr_shims = []          # weak references to the SuperResolution shims of this process, newest last


def enabled() -> bool:
    return os.environ.get('RY_FUSE_STEPS', '1') != '0'


def register_sr(sr) -> None:
    _sr_shims[:] = [r for r in _sr_shims if r() is not None and r() is not sr]
    _sr_shims.append(weakref.ref(sr))


def _partner_sr(device: int):
    for r in reversed(_sr_shims):
        sr = r()
        if sr is not None and sr.device() == device:
            return sr
    return None


class Link(object):
    """One (AcousticConverter shim, SuperResolution shim) pair in one process: the VcCore over their two predictors."""

    def __init__(self, ac, sr):
        mtx = ac.mc2sp_matrix()
        self.sr_ref = weakref.ref(sr)
        self.sr_net = sr._get_net(mtx.shape[1])
        self.core = engine.VcCore(ac._get_net(), self.sr_net, mtx)
        self.pid = os.getpid()
        self.generation = 0

    def alive(self, sr=None) -> bool:
        s = self.sr_ref()
        return (self.pid == os.getpid() and s is not None and s._net is self.sr_net and self.core.handle is not None
                and self.core.stage1.handle is not None and self.sr_net.handle is not None and (sr is None or s is sr))

    def close(self):
        self.core.close()


def link_for(ac):
    """The Link of this AcousticConverter shim with the newest SuperResolution shim on its GPU (None: nothing to fuse with)."""
    if not enabled() or not ac.fusable():
        return None
    link = getattr(ac, '_link', None)
    sr = _partner_sr(ac.device())
    if link is not None and link.alive(sr):
        return link
    if link is not None:
        link.close()
        ac._link = None
    if sr is None:
        return None
    ac._link = Link(ac, sr)
    return ac._link


class Token(object):
    """Says: the rows `convert` produced for THIS feature are still in the device buffer of `link`."""
    __slots__ = ('link', 'generation', 'rows', 'effective')

    def __init__(self, link, rows):
        self.link, self.generation, self.rows, self.effective = link, link.generation, rows, None

    def current(self) -> bool:
        return self.link.alive() and self.link.generation == self.generation


class LazySpectrogram(object):
    """`decode_spectrogram(feature).sp` before anybody has looked at it: exp(mc @ M) + floor, still on the device."""

    def __init__(self, token: Token, mc: numpy.ndarray, alpha: float, fftlen: int, bins: int):
        self._token, self._mc, self._alpha, self._fftlen = token, mc, alpha, fftlen
        self._floor = 0.0
        self._dtype = numpy.dtype(numpy.float64)            # pysptk.mc2sp returns float64
        self.shape = (mc.shape[0], bins)
        self.ndim = 2
        self._array = None

    # ---- what the reference does to it between decode_spectrogram and SuperResolution.convert
    def __iadd__(self, v):
        if numpy.isscalar(v) and self._array is None:
            self._floor += float(v)
            return self
        return self.materialize().__iadd__(v)

    def astype(self, dtype, *a, **k):
        if self._array is not None:
            return self._array.astype(dtype, *a, **k)
        c = LazySpectrogram(self._token, self._mc, self._alpha, self._fftlen, self.shape[1])
        c._floor, c._dtype = self._floor, numpy.dtype(dtype)
        return c

    @property
    def dtype(self):
        return self._dtype

    def __len__(self):
        return self.shape[0]

    # ---- the fused continuation
    def convert_with(self, sr, discard=(0, 0)):
        """`sr.convert(self)` on the device when this object was built for that shim and its rows are still there; else None.
        discard: frames the caller throws away (RY_SR_DISCARD of the shim): not computed, zeros."""
        t = self._token
        if self._array is None and t.current() and t.link.alive(sr):
            core = t.link.core
            if core.discard != tuple(discard):
                core.set_discard(*discard)
            return core.stage2_from_mc(t.effective, self._floor)
        return None

    # ---- everything else sees the array
    def materialize(self) -> numpy.ndarray:
        if self._array is None:
            t = self._token
            if t.current():
                a = t.link.core.mid_sp(t.effective, self._floor)
            else:                                             # the device rows are gone: the host formula on the host copy of mc
                a = sptk.mc2sp_fast(self._mc, self._alpha, self._fftlen) + self._floor
            self._array = numpy.asarray(a, dtype=self._dtype)
        return self._array

    def __array__(self, dtype=None, copy=None):
        a = self.materialize()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __getitem__(self, idx):
        return self.materialize()[idx]

    def __getattr__(self, name):                              # any other ndarray attribute / method
        if name.startswith('_'):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    def __add__(self, v):
        return self.materialize() + v

    __radd__ = __add__

    def __mul__(self, v):
        return self.materialize() * v

    __rmul__ = __mul__

    def __sub__(self, v):
        return self.materialize() - v

    def __truediv__(self, v):
        return self.materialize() / v
