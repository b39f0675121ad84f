#!/usr/bin/env python3
"""Window lanes on their own XCDs (ry_vc_set_lane_xcds): the chained step by lane count, with and without the compute-unit masks, and the
check of the rule the masks rely on (bit i of a CU mask belongs to XCD i % 8).  usage (GPU box): python scripts/gpu_r3_xcdlanes.py [frames] [steps]"""
import ctypes
import os
import sys
import time
from pathlib import Path

import numpy

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault('GPU_MAX_HW_QUEUES', os.environ.get('RY_QUEUES', '24'))
from realtime_yukarin_amd import engine, sptk, synth                # noqa: E402
from realtime_yukarin_amd.weights import flatten_params             # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 160
(d1, P1), (d2, P2) = synth.model_params('SYN-64')
ctx = engine.get_context(0)
# ---- the mask rule
for name, bits in (('none', None), ('i%8==0', [i for i in range(256) if i % 8 == 0]), ('i%8==3', [i for i in range(256) if i % 8 == 3]),
                   ('0..31', list(range(32))), ('32..63', list(range(32, 64))), ('0..127', list(range(128))), ('bit 0', [0]), ('bit 1', [1]), ('bit 8', [8]),
                   ('bit 32', [32]), ('i%32==0', [i for i in range(256) if i % 32 == 0]), ('0..7', list(range(8)))):
    mask = (ctypes.c_uint * 8)()
    if bits:
        for b in bits:
            mask[b >> 5] |= 1 << (b & 31)
    hist = (ctypes.c_uint * 8)()
    ctx.lib.check(ctx.lib.dll.ry_debug_xcc_histogram(ctx.handle, mask if bits else None, 8 if bits else 0, hist))
    print('CU mask %-8s (%3d bits) -> distinct CUs used per XCD %s = %d' % (name, len(bits or []), list(hist), sum(hist)), flush=True)
if os.environ.get('MASK_ONLY'):
    sys.exit(0)

n1 = engine.Net(ctx, d1, flatten_params(d1, P1))
n2 = engine.Net(ctx, d2, flatten_params(d2, P2), width=synth.FFT_BINS - 1)
mtx = sptk.mc2sp_matrix(8, sptk.mcepalpha(16000), 1024)
d_x = ctx.dev_alloc(N * 9); ctx.dev_upload(d_x, synth.stage1_input(N)[0])
d_rows = ctx.dev_alloc(N); ctx.dev_upload(d_rows, numpy.arange(N, dtype=numpy.int32))
d_mc = [ctx.dev_alloc(N * 9) for _ in range(16)]
d_sp = [ctx.dev_alloc(N * 513) for _ in range(16)]
ref = None


def run(lanes, xcd):
    global ref
    core = engine.VcCore(n1, n2, mtx, lanes=lanes)
    if xcd:
        core.set_lane_xcds(True)
    ring = core.ring
    k = [0]

    def step():
        core.enqueue_device(d_x, d_rows, N, N, d_mc[k[0] % ring], d_sp[k[0] % ring], 1e-16)
        k[0] += 1
    t0 = time.perf_counter()
    for _ in range(3 * ring):
        step()
    ctx.sync()
    t_prime = time.perf_counter() - t0
    best = 1e9
    for _ in range(3):
        for _ in range(ring):
            step()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            step()
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) / STEPS * 1e3)
    sp = numpy.empty((N, 513), numpy.float32); ctx.dev_download(d_sp[(k[0] - 1) % ring], sp)
    if ref is None:
        ref = sp
    t0 = time.perf_counter(); step(); ctx.sync(); lat = (time.perf_counter() - t0) * 1e3
    print('lanes %d %-22s %.4f ms per window = %7.0f frames/s   (one window alone on a lane: %.3f ms; priming %.1f s; max rel diff vs the 2-lane result %.1e)' % (
        lanes, ('on %d CUs each' % (256 // lanes)) if xcd else 'whole chip', best, N / (best * 1e-3), lat, t_prime, float(numpy.abs(sp / ref - 1).max())), flush=True)
    core.close()


for lanes, xcd in ((2, False), (2, True), (4, False), (4, True), (8, False), (8, True), (3, False), (6, False), (2, False), (8, True), (8, False), (4, True)):
    run(lanes, xcd)
n1.close(); n2.close()
