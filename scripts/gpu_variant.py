#!/usr/bin/env python3
"""Round 6: an experiment build of the library (libry355<suffix>.so, build.build_product(defs=[...], suffix=...)) against fixed Winograd plans: us of the
Winograd launches (HIP events inside the eager window forward), one line per build.   usage (GPU box): python scripts/gpu_variant.py <suffix | -> <frames> <RY_WINO spec> ..."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
from realtime_yukarin_amd import _lib, engine, synth                # noqa: E402
from realtime_yukarin_amd.weights import flatten_params             # noqa: E402

SUF = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != '-' else ''
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
SPECS = sys.argv[3:] or ['12:1:2:3,13:1:2:3,14:1:2:1,1:1:2:1,3:1:2:5']
NAMES = ['encoder/c%d' % i for i in range(8)] + ['decoder/c%d' % i for i in range(8)]
(d1, P1), (d2, P2) = synth.model_params('SYN-64')
ctx = engine.Context(0, _lib.Ry355Lib(ROOT / 'realtime_yukarin_amd' / ('libry355%s.so' % SUF)))
n2 = engine.Net(ctx, d2, flatten_params(d2, P2), width=synth.FFT_BINS - 1)
out = []
for spec in SPECS:
    os.environ['RY_WINO'] = spec; ctx.reload_env(); n2.set_dtype('f32')
    n2.profile(1, N, 2, window=True)
    t = {q['layer']: (q['ms'] * 1e3, q['grid'][0]) for q in n2.profile(1, N, 10, window=True) if q['name'].startswith('ry_wino')}
    out.append('  '.join('%s %d %6.1f' % (k.replace('encoder/', 'e').replace('decoder/', 'd'), t[k][1], t[k][0]) for k in NAMES if k in t and ('%d:' % NAMES.index(k)) in spec))
print('var %-8s | %s' % (SUF or '-', ' | '.join(out)), flush=True)
