#!/usr/bin/env python3
"""Round 6: compile-time ablations of ry_wino_ldsdma's K loop (-DRY_WINO_ABL=<bits> builds next to the product, libry355_abl<bits>.so: 1 no input transform,
2 no patch fragment reads, 4 no filter fragment reads, 8 no barrier in the K loop, 32 no MFMAs; WRONG results, timing only): us of the Winograd launches under
fixed plans, HIP events inside the eager window forward.       usage (GPU box): python scripts/gpu_r6_abl.py <bits> [frames]"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
from realtime_yukarin_amd import _lib, engine, synth                # noqa: E402
from realtime_yukarin_amd.weights import flatten_params             # noqa: E402

BITS = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
NAMES = ['encoder/c%d' % i for i in range(8)] + ['decoder/c%d' % i for i in range(8)]
lib = ROOT / 'realtime_yukarin_amd' / ('libry355_abl%d.so' % BITS if BITS else 'libry355.so')
(d1, P1), (d2, P2) = synth.model_params('SYN-64')
ctx = engine.Context(0, _lib.Ry355Lib(lib))
n2 = engine.Net(ctx, d2, flatten_params(d2, P2), width=synth.FFT_BINS - 1)
out = []
for spec in ('12:1:2:1,13:1:2:1,14:1:2:1,1:1:2:1,3:1:2:1', '12:1:2:2,13:1:2:3,14:2:2:1,1:2:1:1,3:1:2:5'):
    os.environ['RY_WINO'] = spec; ctx.reload_env(); n2.set_dtype('f32')
    n2.profile(1, N, 2, window=True)
    t = {q['layer']: (q['ms'] * 1e3, q['grid'][0]) for q in n2.profile(1, N, 10, window=True) if q['name'].startswith('ry_wino')}
    out.append('  '.join('%s %d %6.1f' % (NAMES[l].replace('encoder/', 'e').replace('decoder/', 'd'), t[NAMES[l]][1], t[NAMES[l]][0]) for l in (12, 13, 14, 1, 3)))
print('abl %2d | %s | %s' % (BITS, out[0], out[1]), flush=True)
