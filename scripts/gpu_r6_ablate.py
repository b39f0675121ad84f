#!/usr/bin/env python3
"""Round 6: where a Winograd launch spends its time -- the ablation bits of ry_wino_ldsdma (RY_IGEMM_DBG: 8 no K loop = prologue + epilogue only,
4 no output stores, 128 no DMA in the K loop = fragment reads + transforms + MFMAs on stale LDS; WRONG results, timing only) on the planner's plans,
per layer, from HIP events inside the eager window forward.       usage (GPU box): python scripts/gpu_r6_ablate.py [frames] [out]"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
from realtime_yukarin_amd import engine, synth                      # noqa: E402
from realtime_yukarin_amd.weights import flatten_params             # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
OUT = sys.argv[2] if len(sys.argv) > 2 else str(ROOT / 'gpurun_out' / ('r6_ablate_n%d.txt' % N))
NAMES = ['encoder/c%d' % i for i in range(8)] + ['decoder/c%d' % i for i in range(8)]
LAYERS = (1, 2, 3, 4, 11, 12, 13, 14)
(d1, P1), (d2, P2) = synth.model_params('SYN-64')
ctx = engine.get_context(0)
n2 = engine.Net(ctx, d2, flatten_params(d2, P2), width=synth.FFT_BINS - 1)
lines = []


def say(s):
    lines.append(s + '\n'); print(s, flush=True)
    open(OUT, 'w').writelines(lines)


def layer_us(dbg, wino='1', spec=''):
    os.environ['RY_IGEMM_DBG'] = str(dbg); os.environ['RY_WINOGRAD'] = wino
    if spec:
        os.environ['RY_WINO'] = spec
    else:
        os.environ.pop('RY_WINO', None)
    ctx.reload_env(); n2.set_dtype('f32')
    n2.profile(1, N, 2, window=True)
    out = {}
    for q in n2.profile(1, N, 10, window=True):
        if q['name'].startswith(('ry_wino', 'ry_igemm')):
            out[q['layer']] = (q['ms'] * 1e3, q['name'], q['grid'][0])
    return out


say('# ablations of the MFMA-bound stage-2 launches (the GEMM launch alone, no reduce / copy nodes), SYN-64, %d frames; us' % N)
for wino, spec in (('1', ''), ('1', os.environ.get('ABLATE_SPEC', '12:1:2:3,13:1:2:3')), ('0', '')):
    tabs = {d: layer_us(d, wino, spec) for d in (0, 8, 12, 4, 128)}
    say('# RY_WINOGRAD=%s RY_WINO=%s' % (wino, spec))
    say('%-11s %-40s %6s %9s %9s %9s %9s %9s' % ('layer', 'kernel', 'grid', 'full', 'no-K', 'no-K/st', 'no-store', 'no-DMA'))
    for l in LAYERS:
        nm = NAMES[l]
        say('%-11s %-40s %6d %9.2f %9.2f %9.2f %9.2f %9.2f' % (nm, tabs[0][nm][1], tabs[0][nm][2], tabs[0][nm][0], tabs[8][nm][0], tabs[12][nm][0], tabs[4][nm][0], tabs[128][nm][0]))
os.environ['RY_IGEMM_DBG'] = '0'
