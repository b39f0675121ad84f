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
    c = _lib.RyNetDesc(desc.ndim, desc.in_ch, desc.out_ch, desc.base, desc.extensive_layers, 512 if desc.ndim == 2 else 1, 2e-5, 0.2, int(desc.glu))
    assert int(lib.dll.ry_net_param_count(ctypes.byref(c))) == param_count(desc)
    assert param_count(desc) == sum(int(numpy.prod(s)) for _, s in param_list(desc))


def test_bad_descriptor_is_an_error_not_a_crash(lib):
    c = _lib.RyNetDesc(3, 9, 9, 64, 8, 1, 2e-5, 0.2, 0)
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


TILE_DIMS = {1: (128, 128), 2: (256, 64), 3: (64, 128), 4: (32, 128), 5: (128, 64), 6: (96, 128), 7: (256, 128)}


@pytest.mark.parametrize('frames', [128, 384, 1024])
@pytest.mark.parametrize('batch', [1, 4])
def test_stage2_planner_picks_legal_launches(lib, frames, batch):
    """Host logic of the stage-2 planner (no device work): for every implicit-GEMM layer shape of the base-64 predictor the
    chosen tile divides Cout, split-K x K groups never exceeds the K chunks, K groups only with the 4-wave tiles."""
    ch = [64, 128, 256, 512, 512, 512, 512, 512]
    shapes = []
    h, w = frames, 512
    for i in range(1, 8):                                  # encoder c1..c7: k4 s2 convolutions
        h, w = h // 2, w // 2
        shapes.append((batch * h * w, ch[i], 1, 16 * ch[i - 1] // 32))
    dec_in, dec_out = [512, 1024, 1024, 1024, 1024, 512, 256], [512, 512, 512, 512, 256, 128, 64]
    for j in range(7):                                     # decoder c0..c6: k4 s2 deconvolutions = 4 phases of 2x2 taps
        shapes.append((batch * h * w, dec_out[j], 4, 4 * dec_in[j] // 32))
        h, w = h * 2, w * 2
    for M, N, nph, nk in shapes:
        t, s, k, e = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
        lib.check(lib.dll.ry_debug_plan_igemm(M, N, nph, nk, ctypes.byref(t), ctypes.byref(s), ctypes.byref(k), ctypes.byref(e)))
        bm, bn = TILE_DIMS[t.value]
        assert N % bn == 0
        assert 1 <= s.value and k.value in (1, 2) and s.value * k.value <= nk
        assert k.value == 1 or bm <= 128
        assert e.value > 0.0


@pytest.mark.parametrize('frames', [128, 384, 1024])
@pytest.mark.parametrize('mode', [1, 2])
def test_stage2_planner_picks_legal_launches_bf16(lib, frames, mode):
    """The same for the bf16 (mode 1) and split-bf16 (mode 2: K three times as long) kernels: 64-channel chunks, tiles up to
    128 rows only, and -- measured on MI355X -- the split-bf16 plans of the MFMA-bound layers keep one K group per workgroup."""
    ch = [64, 128, 256, 512, 512, 512, 512, 512]
    shapes = []
    h, w = frames, 512
    kmul = 3 if mode == 2 else 1
    for i in range(1, 8):
        h, w = h // 2, w // 2
        shapes.append((h * w, ch[i], 1, kmul * 16 * ch[i - 1] // 64))
    dec_in, dec_out = [512, 1024, 1024, 1024, 1024, 512, 256], [512, 512, 512, 512, 256, 128, 64]
    for j in range(7):
        shapes.append((h * w, dec_out[j], 4, kmul * 4 * dec_in[j] // 64))
        h, w = h * 2, w * 2
    f = lib.dll.ry_debug_plan_igemm_bf16
    for M, N, nph, nk in shapes:
        t, s, k, e = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
        lib.check(f(mode, M, N, nph, nk, ctypes.byref(t), ctypes.byref(s), ctypes.byref(k), ctypes.byref(e)))
        bm, bn = TILE_DIMS[t.value]
        assert N % bn == 0 and bm <= 128
        assert 1 <= s.value and k.value in (1, 2) and s.value * k.value <= nk
        assert e.value > 0.0
        t2, s2, k2 = ctypes.c_int(t.value), ctypes.c_int(s.value), ctypes.c_int(k.value)       # a given plan is kept as it is
        lib.check(f(mode, M, N, nph, nk, ctypes.byref(t2), ctypes.byref(s2), ctypes.byref(k2), None))
        assert (t2.value, s2.value, k2.value) == (t.value, s.value, k.value)
    assert f(3, 100, 128, 1, 8, ctypes.byref(t), ctypes.byref(s), ctypes.byref(k), None) != 0


def test_output_stationary_planner_follows_the_measured_ranking(lib):
    """choose_os2 (round 5): the slice cost fitted to the MI355X sweeps (profiles/r05/e_os_sweep_*) picks the measured winners of the bottom
    layers of SYN-64 -- (rows per phase, channels, phases, K units) -> (tile rows / 4, tile channels / 4) -- and keeps encoder c5 at 300 frames
    and decoder c2 at 100 frames (cost x units 10240 / 8192) on the implicit GEMM."""
    f = lib.dll.ry_debug_plan_os2
    picks = {}
    for name, (M, N, nph, U) in dict(e7_300=(12, 512, 1, 128), e6_300=(48, 512, 1, 128), d0_300=(12, 512, 4, 32), d1_300=(48, 512, 4, 64), e5_300=(192, 512, 1, 128),
                                     e7_100=(4, 512, 1, 128), e6_100=(16, 512, 1, 128), e5_100=(64, 512, 1, 128), d0_100=(4, 512, 4, 32), d1_100=(16, 512, 4, 64),
                                     d2_100=(64, 512, 4, 64)).items():
        a, b, w, d, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
        lib.check(f(M, N, nph, U, ctypes.byref(a), ctypes.byref(b), ctypes.byref(w), ctypes.byref(d), ctypes.byref(c)))
        assert U % (4 * w.value) == 0 and d.value in (2, 4) and N % (4 * b.value) == 0
        picks[name] = (a.value, b.value, c.value * U)
    assert picks['e7_300'][:2] == (1, 2) and picks['e6_300'][:2] in ((2, 4), (3, 2)) and picks['d1_300'][:2] == (6, 4), picks
    assert picks['e7_100'][:2] == (1, 1) and picks['e6_100'][:2] == (1, 2) and picks['e5_100'][:2] == (2, 4), picks
    assert all(picks[k][2] <= 4608 for k in picks if k not in ('e5_300', 'd2_100')) and picks['e5_300'][2] > 4608 and picks['d2_100'][2] > 4608, picks
    a, b, w, d = ctypes.c_int(3), ctypes.c_int(1), ctypes.c_int(8), ctypes.c_int(2)                  # a given slice is kept
    lib.check(f(12, 512, 1, 128, ctypes.byref(a), ctypes.byref(b), ctypes.byref(w), ctypes.byref(d), None))
    assert (a.value, b.value, w.value, d.value) == (3, 1, 8, 2)
    a, b, w, d = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert f(12, 512, 1, 6, ctypes.byref(a), ctypes.byref(b), ctypes.byref(w), ctypes.byref(d), None) != 0      # 6 units: no whole rounds of four per wave


def test_winograd_planner_prefers_short_tiles_and_refuses_grids_no_tile_divides(lib):
    """choose_wino (round 6): (stencil grid per phase, channels, phases, patches of 16 input channels) -> (workgroup shape, M-blocks per tile row, split).
    Short tiles win where both divide the grid -- the dead-row crop and the copied padding rows round to whole tile rows (8 x 32 against 16 x 16 pixels:
    two-lane step -1.1 % encoder, -2.8 % decoder, profiles/r06/plan_ab_mbw.txt); decoder c5 of the 300-frame window stays unsplit (a 3-way split is 25 us
    faster alone and 1.5 - 3.7 % slower under two lanes, profiles/r06/wino_lanes_sweep_n300.txt); the 12 x 16 grids of encoder c5 / decoder c2 have no tile."""
    def plan(Mh, Mw, N, nph, npat, B=1, cfg=0, mbw=0, sp=0):
        c, m, s_ = ctypes.c_int(cfg), ctypes.c_int(mbw), ctypes.c_int(sp)
        rc = lib.dll.ry_debug_plan_wino(Mh, Mw, N, nph, npat, B, ctypes.byref(c), ctypes.byref(m), ctypes.byref(s_))
        return rc, c.value, m.value, s_.value
    # SYN-64 at 384 padded frames: encoder c1 .. c4 (convolutions: 4 parities x Cin / 16 patches), decoder c3 .. c6 (4 phases)
    layers = dict(e1=(192, 256, 128, 1, 16), e2=(96, 128, 256, 1, 32), e3=(48, 64, 512, 1, 64), e4=(24, 32, 1024, 1, 128),
                  d3=(24, 32, 512, 4, 64), d4=(48, 64, 256, 4, 64), d5=(96, 128, 128, 4, 32), d6=(192, 256, 64, 4, 16))
    got = {k: plan(*v) for k, v in layers.items()}
    assert all(rc == 0 for rc, *_ in got.values()), got
    for k, (rc, cfg, mbw, sp) in got.items():
        th = 8 * ((2 if cfg == 1 else 4) // mbw)
        assert th == 8, (k, got[k])                                   # 8-row tiles everywhere
        assert 1 <= sp <= layers[k][4]
    assert got['d5'][3] == 1 and got['d6'][3] == 1 and got['e1'][3] == 1          # enough tiles of their own: no slabs
    assert got['e3'][3] > 1 and got['e4'][3] > 1 and got['d3'][3] > 1             # deep K, few tiles: split
    assert plan(12, 16, 1024, 1, 256)[0] != 0 and plan(12, 16, 512, 4, 128)[0] != 0          # encoder c5 / decoder c2: no Winograd tile divides 12 x 16
    assert plan(24, 32, 512, 4, 64, cfg=2)[0] != 0                                # the eight-wave shape has no tile for 24 x 32 either
    rc, cfg, mbw, sp = plan(96, 128, 128, 4, 32, cfg=1, mbw=1, sp=3)              # forced values are kept
    assert (rc, cfg, mbw, sp) == (0, 1, 1, 3)
    assert lib.dll.ry_debug_plan_wino(96, 128, 100, 4, 32, 1, None, None, None) != 0


def test_stage2_planner_near_tie_goes_to_the_small_workgroup(lib):
    """Round 5: where slabs are needed anyway, two K groups per workgroup (125 KiB of LDS: nothing fits beside it on a CU) must beat the external-split-only
    form (62 KiB) by more than 1 % of the estimate.  At 300 / 400 frames that is decoder c3 (768 / 1024 rows per phase, 512 channels, 128 K chunks):
    one K group, external split; encoder c4 / c5 and decoder c2 are 4 - 12 % apart and keep two K groups (the A/B: profiles/r05/r_plan_ab_n300.txt,
    profiles/r05/s_bench_ab.txt; its switch RY_KG_SLABS went with round 6)."""
    def plan(M, N, nph, nk):
        t, s, k, e = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
        lib.check(lib.dll.ry_debug_plan_igemm(M, N, nph, nk, ctypes.byref(t), ctypes.byref(s), ctypes.byref(k), ctypes.byref(e)))
        return t.value, s.value, k.value, e.value
    d3, c4, c5, d2 = (768, 512, 4, 128), (1536, 512, 1, 256), (384, 512, 1, 256), (192, 512, 4, 128)
    got = {n: plan(*sh) for n, sh in (('d3', d3), ('c4', c4), ('c5', c5), ('d2', d2))}
    assert got['d3'][2] == 1 and got['d3'][1] > 1 and got['d3'][1] % 2 == 0, got
    assert plan(1024, 512, 4, 128)[2] == 1                       # the same layer at 400 frames
    assert all(got[n][2] == 2 and got[n][1] > 1 for n in ('c4', 'c5', 'd2')), got


def test_stage2_planner_rejects_non_igemm_shapes(lib):
    t, s, k = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.dll.ry_debug_plan_igemm(100, 48, 1, 8, ctypes.byref(t), ctypes.byref(s), ctypes.byref(k), None) != 0
    assert b'implicit-GEMM' in lib.dll.ry_last_error()
