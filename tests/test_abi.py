"""C-ABI surface (CPU, no compute): the product library loads here, exports every symbol include/ry355.h declares,
agrees with the host netspec on the K-list size, and reports errors instead of aborting."""
import ctypes
import re
from pathlib import Path

import numpy
import pytest

from realtime_yukarin_amd import _lib, build
from realtime_yukarin_amd.netspec import NetDesc, param_count, param_list

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope='module')
def lib():
    return _lib.Ry355Lib(build.build_product())


def test_header_and_binding_list_the_same_symbols():
    text = (ROOT / 'include' / 'ry355.h').read_text()
    declared = set(re.findall(r'\b(ry_[a-z0-9_]+)\s*\(', text))
    declared -= {'ry_ctx', 'ry_net', 'ry_net_desc', 'ry_kernel_stat'}
    assert declared == set(_lib.ABI_SYMBOLS), declared ^ set(_lib.ABI_SYMBOLS)


def test_library_exports_every_symbol(lib):
    for name in _lib.ABI_SYMBOLS:
        assert hasattr(lib.dll, name), name


@pytest.mark.parametrize('desc', [NetDesc(1, 9, 9, 64, 8), NetDesc(1, 523, 9, 16, 8), NetDesc(2, 1, 1, 64, 8),
                                  NetDesc(2, 1, 1, 8, 8), NetDesc(1, 9, 9, 8, 3), NetDesc(1, 9, 9, 8, 0)])
def test_param_count_matches_netspec(lib, desc):
    c = _lib.RyNetDesc(desc.ndim, desc.in_ch, desc.out_ch, desc.base, desc.extensive_layers, 512 if desc.ndim == 2 else 1, 2e-5, 0.2)
    assert int(lib.dll.ry_net_param_count(ctypes.byref(c))) == param_count(desc)
    assert param_count(desc) == sum(int(numpy.prod(s)) for _, s in param_list(desc))


def test_bad_descriptor_is_an_error_not_a_crash(lib):
    c = _lib.RyNetDesc(3, 9, 9, 64, 8, 1, 2e-5, 0.2)
    assert int(lib.dll.ry_net_param_count(ctypes.byref(c))) == 0
    assert b'ndim' in lib.dll.ry_last_error()


def test_no_gpu_means_a_loud_failure_not_a_fallback(lib):
    """On a box without a GPU the product path must refuse to run (there is no CPU path)."""
    from realtime_yukarin_amd import engine
    if lib.device_count() > 0:
        pytest.skip('a GPU is visible here')
    with pytest.raises(_lib.Ry355Error):
        engine.Context(0, lib)


def test_missing_library_is_a_loud_failure(tmp_path):
    with pytest.raises(_lib.Ry355Error) as ei:
        _lib.Ry355Lib(tmp_path / 'libry355.so')
    assert 'no CPU fallback' in str(ei.value)
