"""SURVEY.md 8(f) row 4 -- the WORLD synthesis feed of `RealtimeVocoder.decode`
(/root/reference/realtime_voice_conversion/yukarin_wrapper/vocoder.py:88-117), CPU only.

The reference converts every window to Python lists (`sp.tolist()`: N x 513 float objects, twice) so that world4py's
`cast_2d_list_to_2d_pointer` can build a `double**`, and reads every synthesised sample back with a Python list comprehension
(`[synthesizer.buffer[i] for i in range(buffer_size)]`).  For a 100-frame buffer that is ~100 k Python float objects in and 8 k
ctypes indexings out per call -- more host time than both CNNs take on the GPU.  Here the same pointers are built over contiguous
float64 numpy buffers (one cast each; the row-pointer table is address arithmetic in numpy) and the ring buffer is read with
`numpy.ctypeslib.as_array`: no per-element Python work.  `decode` has the body and the semantics of the reference method (same
world4py calls in the same order, the same `_before_buffer` keep-alive of 16 entries) and can be bound over it:

    from realtime_yukarin_amd import vocoder_feed
    RealtimeVocoder.decode = vocoder_feed.decode

world4py itself is not installable here; tests/test_vocoder_feed.py checks this against the reference method on a recording
stand-in of the world4py API."""
import ctypes

import numpy

_PD = ctypes.POINTER(ctypes.c_double)
_PPD = ctypes.POINTER(_PD)


class Feed(object):
    """f0* / sp** / ap** over contiguous float64 copies of one window; keeps the arrays alive as long as it lives."""

    def __init__(self, f0, sp, ap):
        self.f0 = numpy.ascontiguousarray(numpy.asarray(f0, dtype=numpy.float64).reshape(-1))
        self.sp = numpy.ascontiguousarray(sp, dtype=numpy.float64)
        self.ap = numpy.ascontiguousarray(ap, dtype=numpy.float64)
        self.length = len(self.f0)
        if self.sp.shape[0] != self.length or self.ap.shape[0] != self.length:
            raise ValueError('f0 / sp / ap disagree on the frame count: %d %s %s' % (self.length, self.sp.shape, self.ap.shape))
        self.f0_pointer = self.f0.ctypes.data_as(_PD)
        self._sp_rows, self.sp_pointer = self._rows(self.sp)
        self._ap_rows, self.ap_pointer = self._rows(self.ap)

    @staticmethod
    def _rows(a):
        rows = a.ctypes.data + numpy.arange(a.shape[0], dtype=numpy.uint64) * numpy.uint64(a.strides[0])   # address of every row
        rows = numpy.ascontiguousarray(rows, dtype=numpy.uint64)
        return rows, ctypes.cast(rows.ctypes.data, _PPD)


def read_buffer(synthesizer) -> numpy.ndarray:
    """One synthesised ring-buffer block as a fresh float64 array (the reference: a list comprehension over ctypes indexing)."""
    n = int(synthesizer.buffer_size)
    return numpy.ctypeslib.as_array(ctypes.cast(synthesizer.buffer, _PD), shape=(n,)).copy()


def decode(self, acoustic_feature):
    """Drop-in body of `RealtimeVocoder.decode` (vocoder.py:88-117)."""
    from world4py.native import apidefinitions
    from yukarin import Wave
    assert self._synthesizer is not None
    feed = Feed(acoustic_feature.f0, acoustic_feature.sp, acoustic_feature.ap)
    apidefinitions._AddParameters(feed.f0_pointer, feed.length, feed.sp_pointer, feed.ap_pointer, self._synthesizer)
    ys = []
    while apidefinitions._Synthesis2(self._synthesizer) != 0:
        ys.append(read_buffer(self._synthesizer))
    out_wave = Wave(wave=numpy.concatenate(ys) if ys else numpy.empty(0), sampling_rate=self.out_sampling_rate)
    self._before_buffer.append(feed)                         # for holding memory: WORLD reads the parameters during later Synthesis2 calls
    if len(self._before_buffer) > 16:
        self._before_buffer.pop(0)
    return out_wave
