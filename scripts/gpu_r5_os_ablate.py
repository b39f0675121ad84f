#!/usr/bin/env python3
"""Round 5: what is ry_c2d_os waiting for?  The six bottom layers of the 300-frame window on fixed slices, the kernel ablated through
RY_OS2_DBG (a -DRY_OS2_DBG_BUILD build of the library, realtime_yukarin_amd/libry355_dbg.so, loaded through RY355_LIB; wrong results, timing only): 1 no pixel loads, 2 no filter loads, 4 no MFMAs, 8 no K loop, 16 no offset table, 32 no reduction / stores.
us per layer from HIP events inside the eager window forward (as profiles/*layers.txt) and the stage-2 forward as graph replays.

usage (GPU box): python scripts/gpu_r5_os_ablate.py [frames] [out file]"""
import os
import sys
from pathlib import Path

import numpy

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
from realtime_yukarin_amd import engine, synth                      # noqa: E402
from realtime_yukarin_amd.weights import flatten_params             # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
OUT = sys.argv[2] if len(sys.argv) > 2 else str(ROOT / 'gpurun_out' / ('r5_os_ablate_n%d.txt' % N))
NAMES = ['encoder/c%d' % i for i in range(8)] + ['decoder/c%d' % i for i in range(8)]
CFG = os.environ.get('ABLATE_CFG', '6:3:2:8:2,7:1:2:8:2,8:3:2:8:2,9:3:4:8:2')
(d1, P1), (d2, P2) = synth.model_params('SYN-64')
from realtime_yukarin_amd import _lib                             # noqa: E402
DBG_LIB = ROOT / 'realtime_yukarin_amd' / 'libry355_dbg.so'       # python -c "from realtime_yukarin_amd import build; build.build_product(defs=['RY_OS2_DBG_BUILD=1'], suffix='_dbg')"
ctx = engine.Context(0, _lib.Ry355Lib(DBG_LIB))
n2 = engine.Net(ctx, d2, flatten_params(d2, P2), width=synth.FFT_BINS - 1)
d_in = ctx.dev_alloc(N * 513); d_out = ctx.dev_alloc(N * 513)
ctx.dev_upload(d_in, synth.stage2_input(N)[0])
lines = []


def say(s):
    lines.append(s + '\n')
    print(s, flush=True)


def run(dbg, reps=20):
    os.environ['RY_OS2'] = CFG; os.environ['RY_OS2_DBG'] = str(dbg)
    n2.set_dtype('f32')
    n2.profile(1, N, 2, window=True)
    st = n2.profile(1, N, reps, window=True)
    lu = {}
    for q in st:
        lu[q['layer']] = lu.get(q['layer'], 0.0) + q['ms'] * 1e3
    for _ in range(3):
        n2.convert_device(d_in, d_out, 1, N)
    ctx.sync(); ctx.timer_start()
    for _ in range(30):
        n2.convert_device(d_in, d_out, 1, N)
    return lu, ctx.timer_stop() / 30


say('# ry_c2d_os ablations, SYN-64, %d-frame window, RY_OS2=%s' % (N, CFG))
say('# %-44s %s   forward (graph replay) ms' % ('RY_OS2_DBG', '  '.join('%11s' % NAMES[l] for l in (5, 6, 7, 8, 9, 10))))
for dbg, what in ((0, 'the kernel'), (4, 'no MFMAs'), (1, 'no pixel loads'), (2, 'no filter loads'), (3, 'no loads'), (7, 'no loads, no MFMAs'),
                  (8, 'no K loop'), (8 + 16, 'no K loop, no offset table'), (8 + 16 + 32, 'empty kernel'), (32, 'no reduction / stores'), (1 + 4, 'filter loads only'), (2 + 4, 'pixel loads only')):
    lu, fw = run(dbg)
    say('  %2d %-40s %s   %.4f' % (dbg, what, '  '.join('%8.2f us' % lu[NAMES[l]] for l in (5, 6, 7, 8, 9, 10)), fw))
Path(OUT).parent.mkdir(parents=True, exist_ok=True)
Path(OUT).write_text(''.join(lines))
