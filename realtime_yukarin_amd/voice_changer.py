"""Host-side mirror of the hot-path entry `VoiceChanger.convert_from_acoustic_feature`
(/root/reference/realtime_voice_conversion/yukarin_wrapper/voice_changer.py:9-42): same constructor arguments,
same attribute names (`acoustic_converter`, `super_resolution`, `threshold`, `output_sampling_rate` -- read by
ConvertStream at realtime_voice_conversion/stream/convert_stream.py:15-27), same step order and in-place
`sp += 1e-16` / float32 cast, so that the reference's own class and this one are interchangeable.  Added on top:
`convert_windows`, which runs the two CNNs for several independent windows in one GPU batch (chunk parallelism
inside a GPU)."""
import os
from typing import List, Optional

import numpy

SP_FLOOR = 1e-16            # voice_changer.py:39


class VoiceChanger(object):
    def __init__(self, acoustic_converter, super_resolution, threshold: float = 60, output_sampling_rate: Optional[int] = None) -> None:
        self.acoustic_converter = acoustic_converter
        self.super_resolution = super_resolution
        self.threshold = threshold
        self.output_sampling_rate = (super_resolution.config.dataset.param.voice_param.sample_rate
                                     if output_sampling_rate is None else output_sampling_rate)

    def _stage1(self, f_in):
        ac = self.acoustic_converter
        f_eff, effective = ac.separate_effective(wave=f_in.wave, feature=f_in, threshold=self.threshold)
        f_out = ac.convert(f_eff) if numpy.any(effective) else f_eff          # all-silent windows skip the CNN
        f_out = ac.combine_silent(effective=effective, feature=f_out)
        f_out = ac.decode_spectrogram(f_out)
        f_out.sp += SP_FLOOR
        return f_out

    def _fused_core(self):
        """`engine.VcCore` when both converters are the MI355X shims in their canonical (mc -> mc) configuration."""
        ac, sr = self.acoustic_converter, self.super_resolution
        if not (hasattr(ac, 'fusable') and hasattr(sr, '_get_net') and ac.fusable()):
            return None
        from . import engine
        key = os.getpid()
        mtx = ac.mc2sp_matrix()
        n1, n2 = ac._get_net(), sr._get_net(mtx.shape[1])
        if getattr(self, '_core_pid', None) == key and not self._core.alive_on(n1, n2):
            self._core.close()                                 # a converter was closed / rebuilt (another bin count) under the core: never submit on a freed predictor
            self._core_pid = None
        if getattr(self, '_core_pid', None) != key:
            self._core = engine.VcCore(n1, n2, mtx)
            self._core_pid = key
            if os.environ.get('RY_VC_WARM'):                   # e.g. RY_VC_WARM=300: plans and graphs of every ring slot before the first window
                self._core.warm(int(os.environ['RY_VC_WARM']))
        return self._core

    def close(self) -> None:
        """Free the device-resident core (ring buffers, pinned staging) in the process that built it; the converters stay usable."""
        if getattr(self, '_core_pid', None) == os.getpid() and getattr(self, '_core', None) is not None:
            self._core.close()
        self._core, self._core_pid = None, None

    def convert_from_acoustic_feature(self, f_in, discard=(0, 0)):
        return self.finish(self.begin(f_in, discard))

    # ---- the same call in two halves: `begin` queues the window on the GPU and returns at once (up to six windows may be in flight,
    # `ry_vc_submit` / `ry_vc_submit_wave`), `finish` waits for it and assembles the output feature.  `worker.convert_worker` uses the
    # pair to keep a backlog of windows pipelined: H2D of window i + 1 and D2H of window i - 1 run under the kernels of window i.
    def begin(self, f_in, discard=(0, 0)):
        """discard = (front, back): the caller will throw away that many leading / trailing frames of the result (`ConvertStream.process`
        picks [pad, -pad) of what it converted, convert_stream.py:40-42; `worker.convert_worker` passes its pad).  On the device-resident
        path stage 2 then does not compute them (`ry_vc_set_discard`): their spectrogram rows come back as zeros, every kept row is
        bit-identical to the full result, mc / f0 / ap are always complete.  Ignored on the generic path."""
        core = self._fused_core()
        if core is None:                       # generic path: the reference's step order, one call per step (nothing to overlap)
            f_out = self._stage1(f_in)
            f_out.sp = self.super_resolution.convert(f_out.sp.astype(numpy.float32))
            return ('done', f_out)
        # device-resident path: everything in one window call.  The silence gate runs on the device too (ry_vc_submit_wave) when it
        # can restate the host arithmetic bit for bit (float32 wave, absolute reference, power-of-two frame length); otherwise the
        # mask is taken on the host (it reads the raw wave) and only the effective rows go up (ry_vc_submit).
        discard = (int(discard[0]), int(discard[1])) if os.environ.get('RY_DISCARD_HINT', '1') != '0' else (0, 0)
        if core.discard != discard:
            core.set_discard(*discard)
        ac = self.acoustic_converter
        from . import gate
        from yukarin.wave import default_effective_ref
        param = ac.config.dataset.acoustic_param
        wave = f_in.wave
        thr = self.threshold if self.threshold is not None else param.threshold_db
        w = numpy.asarray(wave.wave)
        if os.environ.get('RY_DEVICE_GATE', '1') != '0' and gate.device_gate_usable(w, param.fft_length, thr, default_effective_ref()):
            hop, _ = wave.get_hop_and_length(param.frame_period)
            p_eff, p_all = gate.thresholds(thr)
            t = core.submit_wave(w, hop, param.fft_length, p_eff, p_all, numpy.asarray(f_in.mc, dtype=numpy.float32), SP_FLOOR)
            return ('wave', core, t, f_in)
        f_eff, effective = ac.separate_effective(wave=wave, feature=f_in, threshold=self.threshold)
        t = core.submit(numpy.asarray(f_eff.mc, dtype=numpy.float32), effective, SP_FLOOR)
        return ('host', core, t, f_eff, effective)

    def finish(self, handle, lean: bool = False):
        """lean=True (the multi-GPU dispatcher): everything of the result except `ap`, which the stage never touches
        (`AcousticConverter.convert` passes it through, combine_silent zeroes the silent frames) -- the caller keeps its own `ap` and needs
        `effective` to finish the job (`attach_ap`).  Same values and dtypes as the full result for f0 / voiced / mc / sp."""
        if handle[0] == 'done':
            f_out = handle[1]
            if lean:
                f_out.effective = None                      # generic path: `ap` came through the converters, nothing to re-attach
            return f_out
        ac = self.acoustic_converter
        if handle[0] == 'wave':
            _, core, t, f_in = handle
            mc, sp, effective = core.wait_wave(t)
            f_eff = f_in.indexing(effective)
        else:
            _, core, t, f_eff, effective = handle
            mc, sp = core.wait(t)
        if lean:
            from yukarin.acoustic_feature import AcousticFeature
            n = len(effective)
            f0, voiced = numpy.zeros((n, 1), numpy.float32), numpy.zeros((n, 1), bool)       # AcousticFeature.silent
            if len(f_eff.f0):
                f0[effective] = ac.f0_converter.convert(f_eff).f0 if ac.f0_converter is not None else f_eff.f0
                voiced[effective] = f_eff.voiced
            f_out = AcousticFeature(f0=f0, voiced=voiced, mc=mc, sp=sp)
            f_out.effective = effective
            return f_out
        f_out = ac.combine_silent(effective=effective, feature=self._passthrough(f_eff))
        f_out.mc = mc
        f_out.sp = sp
        return f_out

    @staticmethod
    def attach_ap(ap_in, effective, first: int, last: int, bins: int):
        """The `ap` rows [first, last) of a window's result from the window's INPUT `ap`: `AcousticConverter.convert` passes ap through and
        `combine_silent` leaves zeros on the frames the silence gate cut (float32, as `AcousticFeature.silent` allocates it)."""
        n = last - first
        if not isinstance(ap_in, numpy.ndarray):
            return numpy.zeros((n, bins), numpy.float32)            # (`bins` = the ap width of the INPUT rate: AcousticConverter._sizes()['ap'])
        ap = numpy.array(ap_in[first:last], dtype=numpy.float32)      # always a fresh array: the result never aliases the submitted window (round-4 advisor)
        if effective is not None and not effective[first:last].all():
            ap[~effective[first:last]] = 0
        return ap

    def _passthrough(self, f_eff):
        """f0 through the F0Converter, ap / voiced untouched -- what `AcousticConverter.convert` returns besides mc."""
        ac = self.acoustic_converter
        from yukarin.acoustic_feature import AcousticFeature
        n = len(f_eff.f0)
        f0 = f_eff.f0
        if n and ac.f0_converter is not None:
            f0 = ac.f0_converter.convert(f_eff).f0
        return AcousticFeature(f0=f0, ap=f_eff.ap, voiced=f_eff.voiced, mc=numpy.zeros((n, ac.desc.out_ch), numpy.float32))

    def convert_windows(self, f_ins: List, discard=(0, 0)) -> List:
        """Independent windows: stage-1 per window (its length is data dependent after the silence split),
        stage-2 for all equal-length windows in one batched GPU call -- everything on the device when both converters are the MI355X shims."""
        core = self._fused_core()
        if core is not None and f_ins and len({len(f.mc) for f in f_ins}) == 1:
            # device-resident: one call for all windows (`ry_vc_enqueue_device_batch`): stage 1 per window or as a batch, the hop between
            # the CNNs on the device, stage 2 as one batch
            ac = self.acoustic_converter
            discard = (int(discard[0]), int(discard[1])) if os.environ.get('RY_DISCARD_HINT', '1') != '0' else (0, 0)
            if core.discard != discard:
                core.set_discard(*discard)
            split = [ac.separate_effective(wave=f.wave, feature=f, threshold=self.threshold) for f in f_ins]
            res = core.convert_batch([(numpy.asarray(f_eff.mc, dtype=numpy.float32), effective) for f_eff, effective in split], SP_FLOOR)
            outs = []
            for (f_eff, effective), (mc, sp) in zip(split, res):
                f_out = ac.combine_silent(effective=effective, feature=self._passthrough(f_eff))
                f_out.mc = mc
                f_out.sp = sp
                outs.append(f_out)
            return outs
        outs = [self._stage1(f) for f in f_ins]
        lengths = {len(o.sp) for o in outs}
        sr = self.super_resolution
        if len(lengths) == 1 and hasattr(sr, '_get_net'):
            sp = numpy.stack([o.sp.astype(numpy.float32) for o in outs])
            res = sr._get_net(sp.shape[2]).convert(sp)
            for o, r in zip(outs, res):
                o.sp = r
        else:
            for o in outs:
                o.sp = sr.convert(o.sp.astype(numpy.float32))
        return outs
