"""`yukarin.AcousticConverter` on the MI355X ([MEM] method bodies; signatures and call sites:
/root/reference/check.py:54-59, realtime_voice_conversion/converter/yukarin_converter.py:40-47,
realtime_voice_conversion/yukarin_wrapper/voice_changer.py:27-38).

`convert` is the stage-1 hot path: encode_feature -> libry355 `ry_ac_convert` (pad 'minimum' -> Predictor -> crop,
all on the GPU) -> decode_feature; f0 goes through the F0Converter, ap / voiced pass through.  The silence split /
merge and mc -> sp stay on the host exactly where the reference has them.  The object is picklable and fork-safe:
the GPU context and the device-resident predictor are created lazily in whichever process first converts
(run.py:69-79 ships these objects to a child Process)."""
import os
from pathlib import Path

import numpy

from realtime_yukarin_amd import engine, fusion, sptk
from realtime_yukarin_amd.netspec import NetDesc
from realtime_yukarin_amd.weights import flatten_params, load_npz

from .acoustic_feature import AcousticFeature, cheaptrick_fft_size
from .config import Config
from .wave import Wave


def _device_of(gpu) -> int:
    """`gpu` is an ordinal or None.  None means "CPU" in the reference (check.py builds the converters without it);
    this implementation has no CPU path, so None selects the device named by RY_DEVICE (default 0)."""
    return int(os.environ.get('RY_DEVICE', '0')) if gpu is None else int(gpu)


class AcousticConverter(object):
    def __init__(self, config: Config, model_path: Path, gpu: int = None, f0_converter=None, out_sampling_rate: int = None) -> None:
        self.config = config
        self.model_path = model_path
        self.gpu = gpu
        self.f0_converter = f0_converter
        self._param = config.dataset.acoustic_param
        self.out_sampling_rate = self._param.sampling_rate if out_sampling_rate is None else out_sampling_rate
        m = config.model
        self.desc = NetDesc(1, m.in_channels, m.out_channels, m.generator_base_channels, m.generator_extensive_layers, glu=bool(m.glu_generator))
        self._params = load_npz(self.desc, model_path)            # strict K-list / shape validation
        self._net = None
        self._net_pid = None
        self._alpha_out = None
        self._link = None

    # ---- pickling / fork safety: never carry a live HIP handle across processes
    def __getstate__(self):
        d = dict(self.__dict__)
        d['_net'] = None
        d['_net_pid'] = None
        d['_link'] = None
        return d

    def device(self) -> int:
        return _device_of(self.gpu)

    # ---- multi-GPU: a process that receives the predictor by one RCCL broadcast needs no host copy of the weights (dispatch.py)
    def without_weights(self) -> 'AcousticConverter':
        """A copy that carries the config but not the host weights: what the dispatcher ships to the worker processes of GPUs 1 .. G-1,
        which get the device-resident predictor from the broadcast (`adopt_net`) instead of unpickling another 54 MB each."""
        import copy
        c = copy.copy(self)
        c.__dict__.update(self.__getstate__())
        c._params = None
        return c

    def adopt_net(self, net: engine.Net) -> None:
        """Use a device-resident predictor that was built elsewhere in THIS process (from a broadcast weight blob)."""
        if net.desc != self.desc:
            raise ValueError('adopt_net: predictor %r does not match the config %r' % (net.desc, self.desc))
        self.close()
        self._net, self._net_pid = net, os.getpid()

    def _get_net(self) -> engine.Net:
        if self._net is None or self._net_pid != os.getpid():
            if self._params is None:
                raise RuntimeError('this AcousticConverter copy carries no weights (without_weights): call adopt_net with the broadcast predictor first')
            ctx = engine.get_context(_device_of(self.gpu))
            self._net = engine.Net(ctx, self.desc, flatten_params(self.desc, self._params))
            self._net_pid = os.getpid()
            self._link = None
        return self._net

    def close(self) -> None:
        """Free the device-resident predictor (and the fused link built on it) in the process that created it."""
        if self._link is not None and self._link.pid == os.getpid():
            self._link.close()
        self._link = None
        if self._net is not None and self._net_pid == os.getpid():
            self._net.close()
        self._net = None

    # ---- feature <-> array
    def _sizes(self):
        return AcousticFeature.get_sizes(sampling_rate=self._param.sampling_rate, order=self._param.order)

    def _encode_feature(self, feature: AcousticFeature) -> numpy.ndarray:
        """(N, C_in): the columns named by config.dataset.in_features, channels-last (no transpose needed)."""
        cols = []
        for t in self.config.dataset.in_features:
            a = numpy.asarray(getattr(feature, t), dtype=numpy.float32)
            cols.append(a.reshape(a.shape[0], -1))
        x = numpy.concatenate(cols, axis=1)
        if x.shape[1] != self.desc.in_ch:
            raise ValueError('in_features %s give %d channels, model.in_channels is %d'
                             % (self.config.dataset.in_features, x.shape[1], self.desc.in_ch))
        return numpy.ascontiguousarray(x)

    def _decode_feature(self, y: numpy.ndarray) -> dict:
        sizes, out, off = self._sizes(), {}, 0
        for t in self.config.dataset.out_features:
            out[t] = y[:, off:off + sizes[t]]
            off += sizes[t]
        if off != y.shape[1]:
            raise ValueError('out_features %s need %d channels, model.out_channels is %d'
                             % (self.config.dataset.out_features, off, y.shape[1]))
        return out

    # ---- the API VoiceChanger drives
    def separate_effective(self, wave: Wave, feature: AcousticFeature, threshold: float = None):
        """(effective-only feature, per-frame bool mask); host side, as in the reference (voice_changer.py:27-31)."""
        n = len(feature.f0)
        if threshold is None:
            threshold = self._param.threshold_db
        if threshold is None:
            return feature, numpy.ones(n, dtype=bool)
        effective = wave.get_effective_frame(threshold_db=threshold, fft_length=self._param.fft_length,
                                             frame_period=self._param.frame_period)
        if len(effective) < n:
            effective = numpy.concatenate([effective, numpy.zeros(n - len(effective), dtype=bool)])
        effective = effective[:n]
        return feature.indexing(effective), effective

    def convert(self, in_feature: AcousticFeature) -> AcousticFeature:
        x = self._encode_feature(in_feature)
        link = fusion.link_for(self) if len(x) else None
        if link is not None:                                        # ry_vc_stage1: the same CNN; the converted rows also stay on the device
            y = link.core.convert_stage1(x)
            link.generation += 1
        else:
            y = self._get_net().convert(x)                          # ry_ac_convert: the stage-1 CNN on the MI355X
        d = self._decode_feature(y)
        f0 = in_feature.f0
        if self.f0_converter is not None:
            f0 = self.f0_converter.convert(in_feature).f0
        out = AcousticFeature(f0=f0, ap=in_feature.ap, voiced=in_feature.voiced)
        for k, v in d.items():
            setattr(out, k, v)
        if link is not None:
            out._ry_token = fusion.Token(link, y.copy())
        return out

    def combine_silent(self, effective: numpy.ndarray, feature: AcousticFeature) -> AcousticFeature:
        silent = AcousticFeature.silent(len(effective), sizes=self._sizes(), keys=('mc', 'ap', 'f0', 'voiced'))
        silent.indexing_set(effective, feature)
        tok = getattr(feature, '_ry_token', None)
        if tok is not None and tok.current() and numpy.array_equal(feature.mc, tok.rows):   # the rows convert() returned, untouched since
            tok.effective = numpy.array(effective, dtype=bool)
            silent._ry_token = tok
            silent._ry_mc = silent.mc
        return silent

    def decode_spectrogram(self, feature: AcousticFeature) -> AcousticFeature:
        if self._alpha_out is None:
            self._alpha_out = sptk.mcepalpha(self.out_sampling_rate)
        fftlen = cheaptrick_fft_size(self.out_sampling_rate)
        tok = getattr(feature, '_ry_token', None)
        if (tok is not None and tok.effective is not None and tok.current() and feature.mc is getattr(feature, '_ry_mc', None)
                and numpy.array_equal(feature.mc[tok.effective], tok.rows) and not feature.mc[~tok.effective].any()):
            # the mc of this feature is still what stage 1 left on the device: the spectrogram is formed there when somebody needs it
            feature.sp = fusion.LazySpectrogram(tok, feature.mc, self._alpha_out, fftlen, fftlen // 2 + 1)
            return feature
        # pysptk.mc2sp == exp(mc @ M): every step before the exp is linear (sptk.mc2sp_matrix); float64 on the host like SPTK
        feature.sp = sptk.mc2sp_fast(numpy.asarray(feature.mc, dtype=numpy.float32), alpha=self._alpha_out, fftlen=fftlen)
        return feature

    def mc2sp_matrix(self) -> numpy.ndarray:
        """The (order+1, bins) matrix of `decode_spectrogram` for this converter's output rate (used by the fused device path)."""
        if self._alpha_out is None:
            self._alpha_out = sptk.mcepalpha(self.out_sampling_rate)
        return sptk.mc2sp_matrix(self.desc.out_ch - 1, self._alpha_out, cheaptrick_fft_size(self.out_sampling_rate))

    def fusable(self) -> bool:
        """True when convert() is exactly mc -> mc (the canonical config), so stage-1 -> mc2sp -> stage-2 can stay on the GPU."""
        return list(self.config.dataset.in_features) == ['mc'] and list(self.config.dataset.out_features) == ['mc']
