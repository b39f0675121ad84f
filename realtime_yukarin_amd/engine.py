"""Host-side handles over the C ABI: one `Context` per (process, GPU), one `Net` per predictor.

Mirrors what the reference's dependencies do around their Chainer models ([MEM]; call sites
/root/reference/realtime_voice_conversion/converter/yukarin_converter.py:40-55): build the predictor
from the config's `model` section, load the npz, move it to the GPU, then run `model(x)` per buffer.
Contexts are created lazily per process id: the reference constructs the converter objects in the
parent and ships them to a child `Process` (/root/reference/run.py:39-46,69-79), and a HIP context
must never cross a fork.
"""
import ctypes
import os
import weakref
from typing import Dict, List, Optional

import numpy

from . import _lib
from .netspec import NetDesc, pad_frames, param_count


class Context(object):
    def __init__(self, device: int = 0, lib: Optional[_lib.Ry355Lib] = None):
        self.lib = lib if lib is not None else _lib.default_lib()
        self.device = int(device)
        self.pid = os.getpid()
        n = self.lib.device_count()
        if n < 1:
            raise _lib.Ry355Error('no HIP device is visible: realtime_yukarin_amd needs an MI355X (there is no CPU path)')
        h = ctypes.c_void_p()
        self.lib.check(self.lib.dll.ry_init(self.device, ctypes.byref(h)))
        self.handle = h

    def sync(self):
        self.lib.check(self.lib.dll.ry_sync(self.handle))

    @property
    def stream(self) -> int:
        return int(self.lib.dll.ry_stream(self.handle) or 0)

    def reload_env(self):
        """Read the process-wide RY_* switches again (`ry_debug_reload_env`; tests and A/B scripts).  Launch plans that exist keep their choices until
        `Net.set_dtype` drops them."""
        self.lib.check(self.lib.dll.ry_debug_reload_env())

    def timer_start(self):
        self.lib.check(self.lib.dll.ry_timer_start(self.handle))

    def timer_stop(self) -> float:
        ms = ctypes.c_float()
        self.lib.check(self.lib.dll.ry_timer_stop(self.handle, ctypes.byref(ms)))
        return float(ms.value)

    def close(self):
        if self.handle is not None and self.pid == os.getpid():
            self.lib.dll.ry_shutdown(self.handle)
        self.handle = None

    def dev_alloc(self, n_floats: int) -> int:
        p = ctypes.POINTER(ctypes.c_float)()
        self.lib.check(self.lib.dll.ry_dev_alloc(self.handle, int(n_floats), ctypes.byref(p)))
        return ctypes.cast(p, ctypes.c_void_p).value

    def dev_free(self, ptr: int):
        self.lib.check(self.lib.dll.ry_dev_free(self.handle, _lib._fptr(int(ptr))))

    def dev_upload(self, ptr: int, a: numpy.ndarray):
        a = numpy.ascontiguousarray(a)
        assert a.dtype.itemsize == 4, 'four-byte elements (float32 / int32)'
        self.lib.check(self.lib.dll.ry_dev_upload(self.handle, _lib._fptr(int(ptr)), ctypes.cast(a.ctypes.data, ctypes.POINTER(ctypes.c_float)), a.size))

    def dev_download(self, ptr: int, a: numpy.ndarray):
        assert a.flags['C_CONTIGUOUS'] and a.dtype.itemsize == 4
        self.lib.check(self.lib.dll.ry_dev_download(self.handle, ctypes.cast(a.ctypes.data, ctypes.POINTER(ctypes.c_float)), _lib._fptr(int(ptr)), a.size))

    def mc2sp(self, mc, mtx, floor: float = 0.0):
        """`decode_spectrogram` on the device: exp(mc (N, M) @ mtx (M, F)) + floor."""
        mc = numpy.ascontiguousarray(mc, dtype=numpy.float32)
        mtx = numpy.ascontiguousarray(mtx, dtype=numpy.float32)
        sp = numpy.empty((mc.shape[0], mtx.shape[1]), dtype=numpy.float32)
        self.lib.check(self.lib.dll.ry_mc2sp(self.handle, _lib._fptr(mc), _lib._fptr(mtx), mc.shape[0], mc.shape[1], mtx.shape[1],
                                             float(floor), _lib._fptr(sp)))
        return sp

    # ---- single operators (Chainer layouts in, channels-last activations) ----
    def conv1d(self, x, W, b=None, bn=None, stride=1, pad=0, dilate=1, transposed=False, act=None, splits=0):
        """x (B, L, Cin) -> (B, Lout, Cout).  W (Cout,Cin,k) or transposed (Cin,Cout,k); bn = (gamma,beta,mean,var)."""
        x = numpy.ascontiguousarray(x, dtype=numpy.float32)
        W = numpy.ascontiguousarray(W, dtype=numpy.float32)
        B, L, Cin = x.shape
        Cout = W.shape[1] if transposed else W.shape[0]
        k = W.shape[2]
        Lout = 2 * L if transposed else (L + 2 * pad - dilate * (k - 1) - 1) // stride + 1
        a = _lib.ACTS[act]
        y = numpy.empty((B, max(Lout, 0), Cout // 2 if a == _lib.ACT_GLU else Cout), dtype=numpy.float32)
        bnv = None if bn is None else numpy.ascontiguousarray(numpy.concatenate([numpy.ravel(v) for v in bn]), dtype=numpy.float32)
        bv = None if b is None else numpy.ascontiguousarray(b, dtype=numpy.float32)
        self.lib.check(self.lib.dll.ry_conv1d(self.handle, _lib._fptr(x), B, L, Cin, _lib._fptr(W), _lib._fptr(bv), _lib._fptr(bnv),
                                              Cout, k, stride, pad, dilate, int(bool(transposed)), a, int(splits), _lib._fptr(y)))
        return y

    def conv2d(self, x, W, b=None, bn=None, stride=1, pad=0, transposed=False, act=None, path='auto', tile=None, splits=0, dilate=1):
        """x (B, H, W, Cin) -> (B, Ho, Wo, Cout).  W (Cout,Cin,k,k) or transposed (Cin,Cout,k,k)."""
        x = numpy.ascontiguousarray(x, dtype=numpy.float32)
        W = numpy.ascontiguousarray(W, dtype=numpy.float32)
        B, H, Wd, Cin = x.shape
        Cout = W.shape[1] if transposed else W.shape[0]
        k = W.shape[2]
        Ho = 2 * H if transposed else (H + 2 * pad - dilate * (k - 1) - 1) // stride + 1
        Wo = 2 * Wd if transposed else (Wd + 2 * pad - dilate * (k - 1) - 1) // stride + 1
        y = numpy.empty((B, max(Ho, 0), max(Wo, 0), Cout), dtype=numpy.float32)
        bnv = None if bn is None else numpy.ascontiguousarray(numpy.concatenate([numpy.ravel(v) for v in bn]), dtype=numpy.float32)
        bv = None if b is None else numpy.ascontiguousarray(b, dtype=numpy.float32)
        pth = {'auto': 0, 'igemm': 1, 'direct': 2, 'first': 3, 'last': 4, 'igemm_bf16': 5, 'igemm_x3': 6, 'os': 7, 'wino': 8}[path]
        if path == 'wino':       # Winograd F(2x2, 2x2) form of a k4 s2 p1 layer: tile = (cfg, mbw): workgroup shape (1: 2x2 waves, 2: 4x2 waves) and M-blocks per tile row; zeros / None = the planner's choice
            c = tuple(tile or ()) + (0, 0)
            tcode = int(c[0]) + 16 * int(c[1])
        elif path == 'os':         # output-stationary weight-streaming kernel: tile = (mt4, nt4, waves, depth), zeros / None = the planner's choice
            c = tuple(tile or ()) + (0, 0, 0, 0)
            tcode = int(c[0]) + 16 * int(c[1]) + 256 * int(c[2]) + 8192 * int(c[3])
        else:
            tcode = _lib.TILES[tile]
        self.lib.check(self.lib.dll.ry_conv2d_dilated(self.handle, _lib._fptr(x), B, H, Wd, Cin, _lib._fptr(W), _lib._fptr(bv), _lib._fptr(bnv),
                                                      Cout, k, stride, pad, int(dilate), int(bool(transposed)), _lib.ACTS[act], pth, tcode,
                                                      int(splits), _lib._fptr(y)))
        return y


class VcCore(object):
    """Device-resident core of `VoiceChanger.convert_from_acoustic_feature`: stage-1 on the effective frames ->
    scatter into the silent block -> mc2sp (exp(mc @ M)) -> + floor -> stage-2, one H2D and one D2H per window (`ry_vc_convert`)."""

    def __init__(self, stage1: 'Net', stage2: 'Net', mtx: numpy.ndarray, lanes: Optional[int] = None):
        self.stage1, self.stage2 = stage1, stage2
        self.lib = stage1.ctx.lib
        mtx = numpy.ascontiguousarray(mtx, dtype=numpy.float32)
        self.M, self.F = mtx.shape
        h = ctypes.c_void_p()
        self.lib.check(self.lib.dll.ry_vc_create(stage1.handle, stage2.handle, _lib._fptr(mtx), self.M, self.F, ctypes.byref(h)))
        self.handle = h
        stage1._dependents.add(self); stage2._dependents.add(self)       # `ry_vc` holds raw pointers to both predictors: they must outlive it (Net.close)
        self._pending = {}
        self.discard = (0, 0)
        # lanes: the six ring slots spread over two pairs of predictor handles (`ry_vc_set_lanes`), so that two windows really run side by side
        # (RY_VC_LANES=1: one stage-2 forward after the other; 3 measured slower than 2)
        self.lanes = int(os.environ.get('RY_VC_LANES', '2')) if lanes is None else int(lanes)
        if self.lanes != 1:
            self.set_lanes(self.lanes)

    def reserve(self, n_frames: int):
        """Size the ring for windows of up to n_frames ahead of time (`ry_vc_reserve_frames`): a stream whose windows grow while others are in
        flight is otherwise refused (the ring cannot be re-allocated under a window)."""
        self.lib.check(self.lib.dll.ry_vc_reserve_frames(self.handle, int(n_frames)))

    def warm(self, n_frames: int, rounds: int = 2):
        """Build the launch plans and capture the graphs of every ring slot for windows of n_frames before the first real window
        arrives (each slot otherwise pays them on its first window: tens of milliseconds, once).  Silence-gated windows of another
        effective length still build their stage-1 plan on first sight."""
        x = numpy.zeros((int(n_frames), self.stage1.desc.in_ch), numpy.float32)
        eff = numpy.ones(int(n_frames), bool)
        for _ in range(6 * int(rounds)):
            self.convert(x, eff)

    def set_discard(self, front: int, back: int):
        """The caller will throw away the first `front` / last `back` frames of every following window (`ry_vc_set_discard`): stage 2
        does not compute them, their spectrogram rows come back as zeros; (0, 0) = everything."""
        self.lib.check(self.lib.dll.ry_vc_set_discard(self.handle, int(front), int(back)))
        self.discard = (int(front), int(back))

    def set_lanes(self, lanes: int):
        self.lib.check(self.lib.dll.ry_vc_set_lanes(self.handle, int(lanes)))
        self.lanes = int(lanes)

    @property
    def ring(self) -> int:
        """Windows that may be in flight: six ring slots up to three lanes, else two per lane."""
        return 6 if self.lanes <= 3 else 2 * self.lanes

    @staticmethod
    def _rows(effective):
        effective = numpy.asarray(effective, dtype=bool)
        return int(effective.size), numpy.ascontiguousarray(numpy.nonzero(effective)[0], dtype=numpy.int32)

    def submit(self, x_eff: numpy.ndarray, effective: numpy.ndarray, sp_floor: float = 1e-16) -> int:
        """Queue one window (pinned ring slot -> H2D -> stage-1 -> ... -> D2H) and return its ticket without waiting; up to six
        windows may be in flight (`ry_vc_submit`)."""
        n, rows = self._rows(effective)
        x_eff = numpy.ascontiguousarray(x_eff, dtype=numpy.float32)
        if len(rows):
            x_eff = x_eff.reshape(len(rows), -1)
        t = ctypes.c_int(-1)
        self.lib.check(self.lib.dll.ry_vc_submit(self.handle, _lib._fptr(x_eff) if len(rows) else _lib._fptr(None),
                                                 rows.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), len(rows), n, float(sp_floor), ctypes.byref(t)))
        self._pending[t.value] = n
        return int(t.value)

    def wait(self, ticket: int):
        """(mc (n_frames, M), sp (n_frames, F)) of a submitted window (`ry_vc_wait`)."""
        n = self._pending.pop(ticket)
        mc = numpy.empty((n, self.M), dtype=numpy.float32)
        sp = numpy.empty((n, self.F), dtype=numpy.float32)
        self.lib.check(self.lib.dll.ry_vc_wait(self.handle, int(ticket), _lib._fptr(mc), _lib._fptr(sp)))
        return mc, sp

    def convert(self, x_eff: numpy.ndarray, effective: numpy.ndarray, sp_floor: float = 1e-16):
        """x_eff (n_eff, in_ch) = features of the effective frames, effective (n_frames,) bool -> (mc (n_frames, M), sp (n_frames, F))."""
        return self.wait(self.submit(x_eff, effective, sp_floor))

    def submit_wave(self, wave: numpy.ndarray, hop: int, fft_length: int, p_effective: float, p_all: float, feat: numpy.ndarray,
                    sp_floor: float = 1e-16) -> int:
        """`ry_vc_submit_wave`: the silence gate on the device.  wave float32 (n_samples,), feat (n_frames, in_ch) = the features of ALL
        frames; the thresholds come from `gate.thresholds(threshold_db)`."""
        wave = numpy.ascontiguousarray(wave, dtype=numpy.float32)
        feat = numpy.ascontiguousarray(feat, dtype=numpy.float32)
        t = ctypes.c_int(-1)
        self.lib.check(self.lib.dll.ry_vc_submit_wave(self.handle, _lib._fptr(wave), wave.size, int(hop), int(fft_length), float(p_effective),
                                                      float(p_all), _lib._fptr(feat), feat.shape[0], float(sp_floor), ctypes.byref(t)))
        self._pending[t.value] = feat.shape[0]
        return int(t.value)

    def gate(self, wave: numpy.ndarray, hop: int, fft_length: int, p_effective: float, p_all: float, feat: numpy.ndarray):
        """`ry_vc_gate`: `separate_effective` alone on the device -> (effective (n,) bool, x_eff (n_eff, in_ch), row_of (n_eff,))."""
        wave = numpy.ascontiguousarray(wave, dtype=numpy.float32)
        feat = numpy.ascontiguousarray(feat, dtype=numpy.float32)
        n = feat.shape[0]
        mask = numpy.empty(n, dtype=numpy.uint8)
        x = numpy.empty((n, feat.shape[1]), dtype=numpy.float32)
        rows = numpy.empty(n, dtype=numpy.int32)
        n_eff = ctypes.c_int(0)
        self.lib.check(self.lib.dll.ry_vc_gate(self.handle, _lib._fptr(wave), wave.size, int(hop), int(fft_length), float(p_effective), float(p_all),
                                               _lib._fptr(feat), n, mask.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), ctypes.byref(n_eff),
                                               _lib._fptr(x), rows.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
        return mask.astype(bool), x[:n_eff.value], rows[:n_eff.value]

    def wait_wave(self, ticket: int):
        """(mc, sp, effective (n_frames,) bool) of a window submitted with `submit_wave`."""
        n = self._pending.pop(ticket)
        mc = numpy.empty((n, self.M), dtype=numpy.float32)
        sp = numpy.empty((n, self.F), dtype=numpy.float32)
        mask = numpy.empty(n, dtype=numpy.uint8)
        n_eff = ctypes.c_int(0)
        self.lib.check(self.lib.dll.ry_vc_wait_wave(self.handle, int(ticket), _lib._fptr(mc), _lib._fptr(sp),
                                                    mask.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), ctypes.byref(n_eff)))
        return mc, sp, mask.astype(bool)

    def convert_stream(self, windows, sp_floor: float = 1e-16, depth: int = 2):
        """Generator over (x_eff, effective) pairs -> (mc, sp) in order, keeping `depth` windows in flight: the copies of one window
        run under the kernels of another."""
        if not 1 <= depth <= self.ring:
            raise ValueError('depth must be 1..%d (six ring slots up to three lanes, else two per lane)' % self.ring)
        tickets = []
        for x_eff, effective in windows:
            tickets.append(self.submit(x_eff, effective, sp_floor))
            if len(tickets) >= depth:
                yield self.wait(tickets.pop(0))
        while tickets:
            yield self.wait(tickets.pop(0))

    def enqueue_device(self, x_ptr: int, rows_ptr: int, n_eff: int, n_frames: int, mc_ptr: int, sp_ptr: int, sp_floor: float = 1e-16):
        """Device pointers in and out, nothing waited for (`ry_vc_enqueue_device`)."""
        self.lib.check(self.lib.dll.ry_vc_enqueue_device(
            self.handle, _lib._fptr(int(x_ptr)), ctypes.cast(ctypes.c_void_p(int(rows_ptr)), ctypes.POINTER(ctypes.c_int)),
            int(n_eff), int(n_frames), float(sp_floor), _lib._fptr(int(mc_ptr)), _lib._fptr(int(sp_ptr))))

    def enqueue_device_batch(self, x_ptr: int, rows_ptr: int, n_eff, n_frames: int, mc_ptr: int, sp_ptr: int, sp_floor: float = 1e-16):
        """`ry_vc_enqueue_device_batch`: len(n_eff) windows of n_frames each, device pointers, nothing waited for."""
        ne = numpy.ascontiguousarray(n_eff, dtype=numpy.int32)
        self.lib.check(self.lib.dll.ry_vc_enqueue_device_batch(
            self.handle, len(ne), _lib._fptr(int(x_ptr)), ctypes.cast(ctypes.c_void_p(int(rows_ptr)), ctypes.POINTER(ctypes.c_int)),
            ne.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), int(n_frames), float(sp_floor), _lib._fptr(int(mc_ptr)), _lib._fptr(int(sp_ptr))))

    def convert_batch(self, windows, sp_floor: float = 1e-16):
        """Host arrays in and out for a list of (x_eff, effective) windows of ONE length: [(mc, sp), ...].  Plain (blocking) copies:
        the call is for throughput on a backlog, not for latency."""
        ctx = self.stage1.ctx
        n, cin = len(windows[0][1]), self.stage1.desc.in_ch
        xs, rows, ne = [], [], []
        for x_eff, effective in windows:
            nn, r = self._rows(effective)
            if nn != n:
                raise ValueError('convert_batch needs windows of one length (got %d and %d frames)' % (n, nn))
            x_eff = numpy.ascontiguousarray(x_eff, dtype=numpy.float32)
            if x_eff.shape[0] != len(r):
                raise ValueError('%d effective rows for a mask with %d set frames' % (x_eff.shape[0], len(r)))
            xs.append(x_eff.reshape(len(r), cin)); rows.append(r); ne.append(len(r))
        W, tot = len(windows), int(sum(ne))
        d_x = ctx.dev_alloc(max(tot, 1) * cin); d_r = ctx.dev_alloc(max(tot, 1)); d_mc = ctx.dev_alloc(W * n * self.M); d_sp = ctx.dev_alloc(W * n * self.F)
        try:
            if tot:
                ctx.dev_upload(d_x, numpy.concatenate(xs))
                ctx.dev_upload(d_r, numpy.concatenate(rows).astype(numpy.int32))
            self.enqueue_device_batch(d_x, d_r, ne, n, d_mc, d_sp, sp_floor)
            ctx.sync()
            mc = numpy.empty((W, n, self.M), numpy.float32); sp = numpy.empty((W, n, self.F), numpy.float32)
            ctx.dev_download(d_mc, mc); ctx.dev_download(d_sp, sp)
            if self.discard != (0, 0):                                   # rows the caller said it throws away were not computed: zeros
                k0 = self.discard[0] if self.discard[0] < n else 0
                k1 = n - self.discard[1] if n - self.discard[1] > k0 else n
                sp[:, :k0] = 0; sp[:, k1:] = 0
        finally:
            for q in (d_x, d_r, d_mc, d_sp):
                ctx.dev_free(q)
        return [(mc[w], sp[w]) for w in range(W)]

    # ---- the chain cut where the reference's own VoiceChanger cuts it (voice_changer.py:33-41)
    def convert_stage1(self, x_eff: numpy.ndarray) -> numpy.ndarray:
        """`AcousticConverter.convert` array part; the converted rows also stay on the device for `stage2_from_mc`."""
        x_eff = numpy.ascontiguousarray(x_eff, dtype=numpy.float32)
        y = numpy.empty((x_eff.shape[0], self.M), dtype=numpy.float32)
        self.lib.check(self.lib.dll.ry_vc_stage1(self.handle, _lib._fptr(x_eff), x_eff.shape[0], _lib._fptr(y)))
        return y

    def stage2_from_mc(self, effective: numpy.ndarray, sp_floor: float) -> numpy.ndarray:
        """combine_silent + decode_spectrogram + floor + SuperResolution.convert from the rows `convert_stage1` left on the device."""
        n, rows = self._rows(effective)
        sp = numpy.empty((n, self.F), dtype=numpy.float32)
        self.lib.check(self.lib.dll.ry_vc_stage2_from_mc(self.handle, rows.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), len(rows), n,
                                                         float(sp_floor), _lib._fptr(sp)))
        return sp

    def mid_sp(self, effective: numpy.ndarray, sp_floor: float) -> numpy.ndarray:
        """The intermediate spectrogram exp(mc @ M) + floor of the rows `convert_stage1` left on the device."""
        n, rows = self._rows(effective)
        sp = numpy.empty((n, self.F), dtype=numpy.float32)
        self.lib.check(self.lib.dll.ry_vc_mid_sp(self.handle, rows.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), len(rows), n,
                                                 float(sp_floor), _lib._fptr(sp)))
        return sp

    def close(self):
        """`ry_vc_destroy` dereferences both predictors (their context, the lane clones): it only runs while both are alive.  `Net.close`
        closes its dependent cores first, so the order the caller closes things in does not matter."""
        if (self.handle is not None and self.stage1.handle is not None and self.stage2.handle is not None
                and self.stage1.ctx.handle is not None and self.stage1.ctx.pid == os.getpid()):
            self.lib.dll.ry_vc_destroy(self.handle)
        self.handle = None
        self.stage1._dependents.discard(self); self.stage2._dependents.discard(self)

    def alive_on(self, stage1: 'Net', stage2: 'Net') -> bool:
        """True while this core is open and still sits on exactly these two (open) predictors."""
        return (self.handle is not None and self.stage1 is stage1 and self.stage2 is stage2
                and stage1.handle is not None and stage2.handle is not None)


_contexts: Dict = {}


def get_context(device: int = 0, lib: Optional[_lib.Ry355Lib] = None) -> Context:
    """Per-(pid, device) context, created on first use in the calling process (fork-safe laziness)."""
    key = (os.getpid(), int(device), id(lib) if lib is not None else 0)
    ctx = _contexts.get(key)
    if ctx is None:
        ctx = Context(device, lib)
        _contexts[key] = ctx
    return ctx


class Net(object):
    """A predictor resident on one GPU.  `blob` = flat float32 weights (netspec K-list order), either a
    host ndarray or (ptr, n) of a device buffer (e.g. the tensor an RCCL broadcast filled)."""

    def __init__(self, ctx: Context, desc: NetDesc, blob, width: int = 512, bn_eps: float = 2e-5, lrelu_slope: float = 0.2):
        self.ctx = ctx
        self.desc = desc
        self.width = int(width) if desc.ndim == 2 else 1
        self.cdesc = _lib.RyNetDesc(desc.ndim, desc.in_ch, desc.out_ch, desc.base, desc.extensive_layers, self.width,
                                    bn_eps, lrelu_slope, int(bool(desc.glu)))
        lib = ctx.lib
        want = int(lib.dll.ry_net_param_count(ctypes.byref(self.cdesc)))
        if want != param_count(desc):
            raise _lib.Ry355Error('library and netspec disagree on the parameter count: %d vs %d' % (want, param_count(desc)))
        h = ctypes.c_void_p()
        if isinstance(blob, tuple):
            ptr, n = blob
            lib.check(lib.dll.ry_net_create(ctx.handle, ctypes.byref(self.cdesc), _lib._fptr(int(ptr)), int(n), 1, ctypes.byref(h)))
        else:
            blob = numpy.ascontiguousarray(blob, dtype=numpy.float32)
            lib.check(lib.dll.ry_net_create(ctx.handle, ctypes.byref(self.cdesc), _lib._fptr(blob), blob.size, 0, ctypes.byref(h)))
        self.handle = h
        self._dependents = weakref.WeakSet()                  # VcCores built on this predictor (they hold raw pointers to it)

    def set_dtype(self, dtype: str):
        """'f32' (exact fp32 MFMA, default), 'bf16' (stage-2 only: bf16 operands, fp32 accumulate -- BASELINE config #5) or
        'bf16x3' (stage-2 only: every fp32 product as three bf16 products hi*hi + lo*hi + hi*lo on the bf16 matrix pipe, fp32
        accumulate -- fp32-class results)."""
        self.ctx.lib.check(self.ctx.lib.dll.ry_net_set_dtype(self.handle, {'f32': 0, 'bf16': 1, 'bf16x3': 2}[dtype]))

    def close(self):
        for core in list(self._dependents):                    # a window core on a freed predictor is a use-after-free in `ry_vc_*`: it goes first
            core.close()
        if self.handle is not None and self.ctx.handle is not None and self.ctx.pid == os.getpid():
            self.ctx.lib.dll.ry_net_destroy(self.handle)
        self.handle = None

    # ---- host-array API ----
    def forward(self, x: numpy.ndarray) -> numpy.ndarray:
        """Raw predictor on a padded block.  stage-1: (B, T, in_ch) -> (B, T, out_ch); stage-2: (B, T, width) -> same."""
        x = numpy.ascontiguousarray(x, dtype=numpy.float32)
        B, T = x.shape[0], x.shape[1]
        cin = self.desc.in_ch if self.desc.ndim == 1 else self.width
        cout = self.desc.out_ch if self.desc.ndim == 1 else self.width
        if x.shape != (B, T, cin):
            raise ValueError('expected (B, T, %d), got %s' % (cin, x.shape))
        y = numpy.empty((B, T, cout), dtype=numpy.float32)
        self.ctx.lib.check(self.ctx.lib.dll.ry_net_forward(self.handle, _lib._fptr(x), _lib._fptr(y), B, T, 0))
        return y

    def convert(self, x: numpy.ndarray, discard=(0, 0)) -> numpy.ndarray:
        """The wrapper arithmetic + predictor: stage-1 `AcousticConverter.convert` array part
        ((B,) N, in_ch) -> ((B,) N, out_ch); stage-2 `SuperResolution.convert` ((B,) N, width+1) -> same.
        discard = (front, back), stage 2 only: the caller will throw away that many leading / trailing frames of every window
        (`ry_sr_convert_rows`): they are not computed and come back as zeros, the others are bit-identical to the full call."""
        x = numpy.ascontiguousarray(x, dtype=numpy.float32)
        squeeze = x.ndim == 2
        if squeeze:
            x = x[numpy.newaxis]
        B, N, C = x.shape
        if self.desc.ndim == 1:
            if C != self.desc.in_ch:
                raise ValueError('stage-1 input needs %d channels, got %d' % (self.desc.in_ch, C))
            y = numpy.empty((B, N, self.desc.out_ch), dtype=numpy.float32)
            fn = self.ctx.lib.dll.ry_ac_convert
        else:
            if C != self.width + 1:
                raise ValueError('stage-2 input needs %d bins, got %d' % (self.width + 1, C))
            y = numpy.empty((B, N, C), dtype=numpy.float32)
            fn = self.ctx.lib.dll.ry_sr_convert
        if self.desc.ndim == 2 and tuple(discard) != (0, 0):
            self.ctx.lib.check(self.ctx.lib.dll.ry_sr_convert_rows(self.handle, _lib._fptr(x), _lib._fptr(y), B, N, int(discard[0]), int(discard[1]), 0))
        else:
            self.ctx.lib.check(fn(self.handle, _lib._fptr(x), _lib._fptr(y), B, N, 0))
        return y[0] if squeeze else y

    # ---- device-pointer API (inputs already resident in HBM; only enqueues on the context stream) ----
    def forward_device(self, x_ptr: int, y_ptr: int, batch: int, frames: int):
        self.ctx.lib.check(self.ctx.lib.dll.ry_net_forward(self.handle, _lib._fptr(int(x_ptr)), _lib._fptr(int(y_ptr)), batch, frames, 1))

    def convert_device(self, x_ptr: int, y_ptr: int, batch: int, n_frames: int):
        fn = self.ctx.lib.dll.ry_ac_convert if self.desc.ndim == 1 else self.ctx.lib.dll.ry_sr_convert
        self.ctx.lib.check(fn(self.handle, _lib._fptr(int(x_ptr)), _lib._fptr(int(y_ptr)), batch, n_frames, 1))

    def profile(self, batch: int, frames: int, reps: int = 5, window: bool = False) -> List[dict]:
        """Per-launch timings of the raw forward at `frames` padded rows, or (window=True) of the convert wrapper on one window
        of `frames` real frames -- what the window call runs."""
        stats = (_lib.RyKernelStat * 96)()
        n = ctypes.c_int()
        if window:
            self.ctx.lib.check(self.ctx.lib.dll.ry_net_profile_window(self.handle, frames, reps, stats, 96, ctypes.byref(n)))
        else:
            self.ctx.lib.check(self.ctx.lib.dll.ry_net_profile(self.handle, batch, frames, reps, stats, 96, ctypes.byref(n)))
        return [dict(name=stats[i].name.decode(), layer=stats[i].layer.decode(), ms=float(stats[i].ms),
                     flops=float(stats[i].flops), bytes=float(stats[i].bytes), grid=tuple(stats[i].grid), flops_exec=float(stats[i].flops_exec))
                for i in range(n.value)]
