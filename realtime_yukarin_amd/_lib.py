"""ctypes binding of libry355.so (C ABI: include/ry355.h).

This is the whole host<->device boundary: plain pointers and sizes, no torch types.  The product
library is `realtime_yukarin_amd/libry355.so`, built in-tree by `__graft_entry__.build()`
(hipcc --offload-arch=gfx950).  There is NO CPU fallback: if the library is missing or no GPU is
visible, every entry point of this package raises.  (tests/ may bind a different path -- the
host-side SIMT emulator build of the same sources -- by constructing `Ry355Lib(path)` explicitly;
nothing in this package ever does.)
"""
import ctypes
import os
from pathlib import Path

import numpy

LIB_NAME = 'libry355.so'
DEFAULT_LIB_PATH = Path(__file__).resolve().parent / LIB_NAME

ACT_NONE, ACT_LRELU, ACT_RELU, ACT_GLU = 0, 1, 2, 3
ACTS = {None: ACT_NONE, 'none': ACT_NONE, 'lrelu': ACT_LRELU, 'relu': ACT_RELU, 'glu': ACT_GLU}
PATH_AUTO, PATH_IGEMM, PATH_DIRECT = 0, 1, 2
TILES = {None: 0, 'auto': 0, '128x128': 1, '64x128': 3, '32x128': 4, '128x64': 5, '96x128': 6}
TILES.update({k + 'k2': v + 16 for k, v in list(TILES.items()) if isinstance(k, str) and k != 'auto'})   # two K groups per workgroup
TILES.update({k + 'k1': v + 32 for k, v in list(TILES.items()) if isinstance(k, str) and k[-2:] != 'k2' and k != 'auto'})           # force one

def ensure_hw_queues() -> None:
    """The window call runs up to two windows side by side on their own pairs of HIP streams (ry_vc_set_lanes): the streams only overlap when
    each has a hardware queue of its own.  ROCm hands out 4 by default and folds further streams onto them (measured on MI355X: with a
    second runtime user in the process two lanes then gain nothing, 1.29 ms per window; with 16 every pair of streams overlaps
    (scripts/gpu_queues.py) and two lanes run at 1.16 ms).  The HIP runtime reads GPU_MAX_HW_QUEUES when it starts, so the variable is set
    when the product library is BOUND (`Ry355Lib.__init__`: the first GPU context of the process, `engine.get_context`) -- not at import:
    importing the package changes nothing.  The entry points (bench.py, the worker processes of `dispatch`) set it themselves before
    anything else.  A caller's own value is respected; a runtime that is already up gets a warning."""
    if 'GPU_MAX_HW_QUEUES' in os.environ:
        return
    import sys
    t = sys.modules.get('torch')
    try:
        late = t is not None and t.cuda.is_initialized()
    except Exception:
        late = False
    if late:
        import warnings
        warnings.warn('realtime_yukarin_amd: the HIP runtime of this process started before GPU_MAX_HW_QUEUES=16 could be set; the two window '
                      'lanes may share hardware queues and serialise (set GPU_MAX_HW_QUEUES=16 in the environment before the first HIP call)')
    os.environ['GPU_MAX_HW_QUEUES'] = '16'


# every symbol include/ry355.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = (
    'ry_init', 'ry_shutdown', 'ry_sync', 'ry_stream', 'ry_device_count', 'ry_last_error',
    'ry_net_param_count', 'ry_net_create', 'ry_net_destroy', 'ry_net_clone', 'ry_net_set_dtype', 'ry_net_forward',
    'ry_ac_convert', 'ry_sr_convert', 'ry_sr_convert_rows', 'ry_conv1d', 'ry_conv2d', 'ry_conv2d_dilated',
    'ry_timer_start', 'ry_timer_stop', 'ry_net_profile', 'ry_net_profile_window', 'ry_debug_plan_igemm', 'ry_debug_reload_env', 'ry_debug_stream_overlap', 'ry_debug_plan_igemm_bf16', 'ry_debug_plan_os2', 'ry_debug_plan_wino',
    'ry_vc_create', 'ry_vc_destroy', 'ry_vc_convert', 'ry_mc2sp',
    'ry_vc_submit', 'ry_vc_set_lanes', 'ry_vc_set_discard', 'ry_vc_wait', 'ry_vc_enqueue_device', 'ry_vc_enqueue_device_batch', 'ry_vc_stage1', 'ry_vc_stage2_from_mc', 'ry_vc_mid_sp', 'ry_vc_reserve_frames',
    'ry_vc_submit_wave', 'ry_vc_wait_wave', 'ry_vc_gate',
    'ry_comm_unique_id', 'ry_comm_init', 'ry_comm_destroy', 'ry_comm_bcast_weights', 'ry_comm_allreduce_max', 'ry_comm_barrier',
    'ry_dev_alloc', 'ry_dev_free', 'ry_dev_upload', 'ry_dev_download',
)


class RyNetDesc(ctypes.Structure):
    _fields_ = [('ndim', ctypes.c_int), ('in_ch', ctypes.c_int), ('out_ch', ctypes.c_int),
                ('base', ctypes.c_int), ('extensive_layers', ctypes.c_int), ('width', ctypes.c_int),
                ('bn_eps', ctypes.c_float), ('lrelu_slope', ctypes.c_float), ('glu', ctypes.c_int)]


class RyKernelStat(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char * 48), ('layer', ctypes.c_char * 24), ('ms', ctypes.c_float),
                ('flops', ctypes.c_double), ('bytes', ctypes.c_double), ('grid', ctypes.c_int * 3), ('flops_exec', ctypes.c_double)]


class Ry355Error(RuntimeError):
    pass


_FP = ctypes.POINTER(ctypes.c_float)
_VP = ctypes.c_void_p


def _fptr(a):
    """float* of a C-contiguous float32 ndarray, or a raw device address (int)."""
    if isinstance(a, int):
        return ctypes.cast(ctypes.c_void_p(a), _FP)
    if a is None:
        return ctypes.cast(ctypes.c_void_p(0), _FP)
    assert isinstance(a, numpy.ndarray) and a.dtype == numpy.float32 and a.flags['C_CONTIGUOUS'], \
        'expected a C-contiguous float32 array'
    return a.ctypes.data_as(_FP)


class Ry355Lib(object):
    def __init__(self, path=None):
        path = Path(path) if path is not None else DEFAULT_LIB_PATH
        if not path.exists():
            raise Ry355Error(
                '%s not found: the MI355X HIP library is not built (run `python -c "import __graft_entry__ as g; '
                'g.build()"` at the repo root). This package has no CPU fallback.' % path)
        self.path = path
        ensure_hw_queues()
        self.dll = ctypes.CDLL(str(path))
        d = self.dll
        d.ry_last_error.restype = ctypes.c_char_p
        d.ry_init.argtypes = [ctypes.c_int, ctypes.POINTER(_VP)]
        d.ry_shutdown.argtypes = [_VP]
        d.ry_shutdown.restype = None
        d.ry_sync.argtypes = [_VP]
        d.ry_stream.argtypes = [_VP]
        d.ry_stream.restype = _VP
        d.ry_device_count.restype = ctypes.c_int
        d.ry_net_param_count.argtypes = [ctypes.POINTER(RyNetDesc)]
        d.ry_net_param_count.restype = ctypes.c_size_t
        d.ry_net_create.argtypes = [_VP, ctypes.POINTER(RyNetDesc), _FP, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(_VP)]
        d.ry_net_destroy.argtypes = [_VP]
        d.ry_net_set_dtype.argtypes = [_VP, ctypes.c_int]
        d.ry_net_destroy.restype = None
        d.ry_net_forward.argtypes = [_VP, _FP, _FP, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        d.ry_ac_convert.argtypes = [_VP, _FP, _FP, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        d.ry_sr_convert.argtypes = [_VP, _FP, _FP, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        d.ry_conv1d.argtypes = [_VP, _FP, ctypes.c_int, ctypes.c_int, ctypes.c_int, _FP, _FP, _FP] + [ctypes.c_int] * 8 + [_FP]
        d.ry_conv2d.argtypes = [_VP, _FP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _FP, _FP, _FP] + [ctypes.c_int] * 9 + [_FP]
        d.ry_conv2d_dilated.argtypes = [_VP, _FP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _FP, _FP, _FP] + [ctypes.c_int] * 10 + [_FP]
        d.ry_vc_create.argtypes = [_VP, _VP, _FP, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_VP)]
        d.ry_vc_destroy.argtypes = [_VP]
        d.ry_vc_destroy.restype = None
        d.ry_vc_convert.argtypes = [_VP, _FP, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.c_float, _FP, _FP]
        d.ry_mc2sp.argtypes = [_VP, _FP, _FP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, _FP]
        _IP = ctypes.POINTER(ctypes.c_int)
        d.ry_vc_submit.argtypes = [_VP, _FP, _IP, ctypes.c_int, ctypes.c_int, ctypes.c_float, _IP]
        d.ry_vc_wait.argtypes = [_VP, ctypes.c_int, _FP, _FP]
        d.ry_vc_set_lanes.argtypes = [_VP, ctypes.c_int]
        d.ry_vc_set_discard.argtypes = [_VP, ctypes.c_int, ctypes.c_int]
        d.ry_sr_convert_rows.argtypes = [_VP, _FP, _FP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        d.ry_debug_stream_overlap.argtypes = [_VP, ctypes.c_int, ctypes.c_int, _FP]
        d.ry_net_clone.argtypes = [_VP, ctypes.POINTER(ctypes.c_void_p)]
        d.ry_vc_enqueue_device.argtypes = [_VP, _FP, _IP, ctypes.c_int, ctypes.c_int, ctypes.c_float, _FP, _FP]
        d.ry_vc_enqueue_device_batch.argtypes = [_VP, ctypes.c_int, _FP, _IP, _IP, ctypes.c_int, ctypes.c_float, _FP, _FP]
        d.ry_vc_stage1.argtypes = [_VP, _FP, ctypes.c_int, _FP]
        d.ry_vc_stage2_from_mc.argtypes = [_VP, _IP, ctypes.c_int, ctypes.c_int, ctypes.c_float, _FP]
        d.ry_vc_mid_sp.argtypes = [_VP, _IP, ctypes.c_int, ctypes.c_int, ctypes.c_float, _FP]
        d.ry_vc_reserve_frames.argtypes = [_VP, ctypes.c_int]
        d.ry_vc_submit_wave.argtypes = [_VP, _FP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, _FP, ctypes.c_int,
                                        ctypes.c_float, _IP]
        d.ry_vc_wait_wave.argtypes = [_VP, ctypes.c_int, _FP, _FP, ctypes.POINTER(ctypes.c_ubyte), _IP]
        d.ry_vc_gate.argtypes = [_VP, _FP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, _FP, ctypes.c_int,
                                 ctypes.POINTER(ctypes.c_ubyte), _IP, _FP, _IP]
        d.ry_comm_unique_id.argtypes = [ctypes.c_char_p]
        d.ry_comm_init.argtypes = [_VP, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_VP)]
        d.ry_comm_destroy.argtypes = [_VP]
        d.ry_comm_destroy.restype = None
        d.ry_comm_bcast_weights.argtypes = [_VP, _FP, ctypes.c_size_t, ctypes.c_int]
        d.ry_comm_allreduce_max.argtypes = [_VP, ctypes.POINTER(ctypes.c_double)]
        d.ry_comm_barrier.argtypes = [_VP]
        d.ry_dev_alloc.argtypes = [_VP, ctypes.c_size_t, ctypes.POINTER(_FP)]
        d.ry_dev_free.argtypes = [_VP, _FP]
        d.ry_dev_upload.argtypes = [_VP, _FP, _FP, ctypes.c_size_t]
        d.ry_dev_download.argtypes = [_VP, _FP, _FP, ctypes.c_size_t]
        d.ry_timer_start.argtypes = [_VP]
        d.ry_timer_stop.argtypes = [_VP, ctypes.POINTER(ctypes.c_float)]
        d.ry_net_profile.argtypes = [_VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(RyKernelStat),
                                     ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        d.ry_net_profile_window.argtypes = [_VP, ctypes.c_int, ctypes.c_int, ctypes.POINTER(RyKernelStat), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]

    def check(self, rc):
        if rc != 0:
            raise Ry355Error('libry355: %s (code %d)' % ((self.dll.ry_last_error() or b'').decode('utf-8', 'replace'), rc))

    def device_count(self):
        return int(self.dll.ry_device_count())


_default_lib = None


def default_lib():
    """The product library (never the emulator)."""
    global _default_lib
    if _default_lib is None:
        _default_lib = Ry355Lib()
    return _default_lib
