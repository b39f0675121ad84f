#!/usr/bin/env python3
"""Do the two window lanes drift into lock-step?  The kernel traces (profiles/r03_*_overlap_lanes2.txt) show no MFMA-bound launch on the chip
for 12 % of the two-lane step: a one-round grid of 512-thread workgroups leaves no room for the other lane's big layer, so the lanes
alternate big layer by big layer -- one layer apart -- and reach the small layers at the bottom / the ends of the U-Net together.
Probe: start the stream with the second window held back by d microseconds (host side, once) and time the following windows."""
import os
import sys
import time
from pathlib import Path

import numpy

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
from realtime_yukarin_amd import engine, sptk, synth                # noqa: E402
from realtime_yukarin_amd.weights import flatten_params             # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 100
(d1, P1), (d2, P2) = synth.model_params('SYN-64')
ctx = engine.get_context(0)
n1 = engine.Net(ctx, d1, flatten_params(d1, P1))
n2 = engine.Net(ctx, d2, flatten_params(d2, P2), width=synth.FFT_BINS - 1)
core = engine.VcCore(n1, n2, sptk.mc2sp_matrix(8, sptk.mcepalpha(16000), 1024), lanes=2)
d_x = ctx.dev_alloc(N * 9); ctx.dev_upload(d_x, synth.stage1_input(N)[0])
d_rows = ctx.dev_alloc(N); ctx.dev_upload(d_rows, numpy.arange(N, dtype=numpy.int32))
d_mc = [ctx.dev_alloc(N * 9) for _ in range(6)]
d_sp = [ctx.dev_alloc(N * 513) for _ in range(6)]
k = [0]


def step():
    core.enqueue_device(d_x, d_rows, N, N, d_mc[k[0] % 6], d_sp[k[0] % 6], 1e-16)
    k[0] += 1


for _ in range(24):
    step()
ctx.sync()
for rep in range(3):
    for delay_us in (0, 150, 300, 450, 600, 750, 900, 1100):
        while k[0] % 6:                                  # start every run on ring slot 0 / lane 0
            step()
        ctx.sync()
        t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        while (time.perf_counter() - t1) * 1e6 < delay_us:
            pass
        ts = time.perf_counter()
        for _ in range(STEPS - 1):
            step()
        ctx.sync()
        te = time.perf_counter()
        print('rep %d  second window held back %4d us: %.4f ms per window over %d windows (whole run incl. the delay: %.4f)' % (
            rep, delay_us, (te - t0) * 1e3 / STEPS, STEPS, (te - t0) * 1e3 / STEPS), flush=True)
core.close(); n1.close(); n2.close()
