"""`become_yukarin.SuperResolution` on the MI355X ([MEM] body; signature and call sites: /root/reference/check.py:60-63,
realtime_voice_conversion/converter/yukarin_converter.py:50-55, yukarin_wrapper/voice_changer.py:41).

`convert(sp)` is the stage-2 hot path, run end to end by `ry_sr_convert`: pad 'minimum' along time, log, drop the
last bin, SRPredictor, edge-pad one bin, exp, crop -- one graph replay on the GPU.  Picklable / fork-safe like
`yukarin.AcousticConverter` (lazy per-process context)."""
import os
from pathlib import Path

import numpy

from realtime_yukarin_amd import engine, fusion
from realtime_yukarin_amd.netspec import NetDesc
from realtime_yukarin_amd.weights import flatten_params, load_npz

from .config.sr_config import SRConfig


class SuperResolution(object):
    def __init__(self, config: SRConfig, model_path: Path, gpu: int = None) -> None:
        self.config = config
        self.model_path = model_path
        self.gpu = gpu
        m = config.model
        self.desc = NetDesc(2, 1, 1, m.generator_base_channels, m.generator_extensive_layers)
        self._params = load_npz(self.desc, model_path)
        self._net = None
        self._net_pid = None
        self._bins = None
        fusion.register_sr(self)

    def __getstate__(self):
        d = dict(self.__dict__)
        d['_net'] = None
        d['_net_pid'] = None
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        fusion.register_sr(self)

    def device(self) -> int:
        return int(os.environ.get('RY_DEVICE', '0')) if self.gpu is None else int(self.gpu)

    def close(self) -> None:
        if self._net is not None and self._net_pid == os.getpid():
            self._net.close()
        self._net = None

    # ---- multi-GPU: see yukarin.AcousticConverter.without_weights / adopt_net
    def without_weights(self) -> 'SuperResolution':
        import copy
        c = copy.copy(self)
        c.__dict__.update(self.__getstate__())
        c._params = None
        fusion.register_sr(c)
        return c

    def adopt_net(self, net: engine.Net) -> None:
        """Use a device-resident predictor built in THIS process from a broadcast weight blob (its width fixes the bin count)."""
        if (net.desc.ndim, net.desc.base, net.desc.extensive_layers) != (2, self.desc.base, self.desc.extensive_layers):
            raise ValueError('adopt_net: predictor %r does not match the config %r' % (net.desc, self.desc))
        self.close()
        dtype = os.environ.get('RY_SR_DTYPE', 'f32')
        if dtype != 'f32':
            net.set_dtype(dtype)
        self._net, self._net_pid, self._bins = net, os.getpid(), net.width + 1

    def _get_net(self, bins: int) -> engine.Net:
        if self._net is None or self._net_pid != os.getpid() or self._bins != bins:
            if self._params is None:
                raise RuntimeError('this SuperResolution copy carries no weights (without_weights) and no adopted predictor for %d bins' % bins)
            self.close()                                       # a predictor for another bin count (or another process) is being replaced: free it
            ctx = engine.get_context(self.device())
            self._net = engine.Net(ctx, self.desc, flatten_params(self.desc, self._params), width=bins - 1)
            # opt-in arithmetic of the MFMA-bound layers, for callers that cannot pass an argument (run.py / check.py unchanged):
            # RY_SR_DTYPE = f32 (default, exact) | bf16x3 (split-bf16, ~2e-6 from fp32: DESIGN.md 4.7) | bf16 (BASELINE config #5)
            dtype = os.environ.get('RY_SR_DTYPE', 'f32')
            if dtype not in ('f32', 'bf16', 'bf16x3'):
                raise ValueError("RY_SR_DTYPE must be 'f32', 'bf16' or 'bf16x3', not %r" % dtype)
            if dtype != 'f32':
                self._net.set_dtype(dtype)
            self._net_pid, self._bins = os.getpid(), bins
        return self._net

    @staticmethod
    def discard_frames():
        """RY_SR_DISCARD="front,back" (opt-in, for run.py / check.py unchanged): the caller throws away that many leading / trailing frames of
        every converted window -- `ConvertStream.process` picks [pad, -pad) with pad = extra_time / frame_period
        (realtime_voice_conversion/stream/convert_stream.py:40-42), so a maintainer who runs with convert_extra_time 0.5 s at 5 ms frames sets
        RY_SR_DISCARD=100,100.  Those rows are then not computed (zeros); every other row is bit-identical.  Default: every frame."""
        v = os.environ.get('RY_SR_DISCARD', '')
        if not v:
            return (0, 0)
        a, b = (int(q) for q in v.split(','))
        if a < 0 or b < 0:
            raise ValueError('RY_SR_DISCARD must be "front,back" with non-negative counts, not %r' % v)
        return (a, b)

    def convert(self, input: numpy.ndarray) -> numpy.ndarray:
        """(N, fft_size/2 + 1) float32 spectrogram -> same shape."""
        discard = self.discard_frames()
        if isinstance(input, fusion.LazySpectrogram):          # the spectrogram is still on the GPU (fusion.py): continue there
            out = input.convert_with(self, discard)
            if out is not None:
                return out
        sp = numpy.ascontiguousarray(numpy.asarray(input), dtype=numpy.float32)
        return self._get_net(sp.shape[1]).convert(sp, discard=discard)
