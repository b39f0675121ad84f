"""A recording stand-in of the world4py API surface `RealtimeVocoder.decode` uses (test infrastructure; WORLD itself is not
installable here): `utils.cast_*` build real ctypes arrays from lists exactly like world4py's helpers, `_AddParameters` reads the
parameters back THROUGH THE POINTERS it is given, `_Synthesis2` emits deterministic blocks computed from what it read -- so two
callers that hand over the same values by different means get the same wave, and a caller that hands over wrong or dangling pointers
does not."""
import ctypes
import sys
import types

import numpy

_PD = ctypes.POINTER(ctypes.c_double)


class WorldSynthesizer(object):
    def __init__(self, buffer_size=64, bins=513):
        self.buffer_size = buffer_size
        self.bins = bins
        self._store = (ctypes.c_double * buffer_size)()
        self.buffer = ctypes.cast(self._store, _PD)
        self.queue = []             # one entry per _AddParameters call: per-frame (f0, sum sp, sum ap) read through the pointers
        self.calls = []


def cast_1d_list_to_1d_pointer(values):
    arr = (ctypes.c_double * len(values))(*values)
    return ctypes.cast(arr, _PD) if False else arr


def cast_2d_list_to_2d_pointer(rows):
    keep = [(ctypes.c_double * len(r))(*r) for r in rows]
    table = (_PD * len(rows))(*[ctypes.cast(k, _PD) for k in keep])
    table._keep = keep
    return table


def _AddParameters(f0, length, sp, ap, synth):
    rec = []
    for i in range(length):
        srow = numpy.ctypeslib.as_array(ctypes.cast(sp[i], _PD), shape=(synth.bins,))
        arow = numpy.ctypeslib.as_array(ctypes.cast(ap[i], _PD), shape=(synth.bins,))
        rec.append((float(f0[i]), float(numpy.dot(srow, numpy.arange(1, synth.bins + 1))), float(arow.sum())))
    synth.queue.append(rec)
    synth.calls.append(('add', length))


def _Synthesis2(synth):
    """One block per 2 queued frames (a stand-in for the hop / buffer arithmetic of WORLD): block = f(frame values)."""
    if not synth.queue or len(synth.queue[0]) < 2:
        if synth.queue:
            synth.queue.pop(0)
        return 0
    a, b = synth.queue[0][0], synth.queue[0][1]
    del synth.queue[0][:2]
    block = numpy.sin(numpy.arange(synth.buffer_size) * 0.1 + a[0] * 1e-3) * a[1] + b[2] * 1e-3 + b[0]
    ctypes.memmove(synth._store, numpy.ascontiguousarray(block, dtype=numpy.float64).ctypes.data, 8 * synth.buffer_size)
    synth.calls.append(('synth',))
    return 1


def install(monkeypatch):
    """`world4py.native.{structures, apidefinitions, utils}` -> this module, for the duration of a test."""
    pkg = types.ModuleType('world4py')
    native = types.ModuleType('world4py.native')
    me = sys.modules[__name__]
    native.structures = me
    native.apidefinitions = me
    native.utils = me
    pkg.native = native
    monkeypatch.setitem(sys.modules, 'world4py', pkg)
    monkeypatch.setitem(sys.modules, 'world4py.native', native)
