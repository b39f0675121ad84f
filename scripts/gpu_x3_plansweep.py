#!/usr/bin/env python3
"""Planner check for the split-bf16 ('bf16x3') mode, in ONE process: every (tile, K groups, external splits) choice of one
stage-2 layer at a time through RY_PLAN (re-read by ry_net_set_dtype), timed with HIP events around the eager launches
(`Net.profile`).  Usage (GPU box): python scripts/gpu_x3_plansweep.py [frames] [out file]"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from realtime_yukarin_amd import engine, synth                      # noqa: E402
from realtime_yukarin_amd.netspec import pad_frames                 # noqa: E402
from realtime_yukarin_amd.weights import flatten_params             # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
OUT = sys.argv[2] if len(sys.argv) > 2 else str(ROOT / 'gpurun_out' / ('x3_plansweep_n%d.txt' % N))
T = N + pad_frames(N)
NAMES = ['encoder/c%d' % i for i in range(8)] + ['decoder/c%d' % i for i in range(8)]
TILES = {1: '128x128', 6: '96x128', 3: '64x128', 5: '128x64', 4: '32x128'}
os.environ['RY_X3_MINM'] = os.environ.get('RY_X3_MINM', '1')        # every implicit-GEMM layer on the split path: the sweep decides

(_, _), (d2, P2) = synth.model_params('SYN-64')
ctx = engine.get_context(0)
net = engine.Net(ctx, d2, flatten_params(d2, P2), width=synth.FFT_BINS - 1)
cout = {i: (min(8, 2 ** i) if i < 8 else [8, 8, 8, 8, 4, 2, 1][i - 8]) * 64 for i in range(15)}


def layer_us(plan, layer, reps=4):
    os.environ['RY_PLAN'] = plan
    net.set_dtype('bf16x3')
    st = net.profile(1, T, reps)
    mine = [q for q in st if q['layer'] == NAMES[layer]]
    return sum(q['ms'] for q in mine) * 1e3, ' + '.join('%s %.1f' % (q['name'].replace('ry_igemm_ldsdma', 'g').replace('ry_splitk_reduce', 'red'), q['ms'] * 1e3) for q in mine), \
        sum(q['ms'] for q in st) * 1e3


lines = []
os.environ['RY_PLAN'] = ''
net.set_dtype('bf16x3')
for _ in range(2):
    base = net.profile(1, T, 4)
lines.append('# split-bf16 plan sweep, SYN-64, %d frames (%d padded); microseconds per layer (GEMM + reduce launches), eager launches\n' % (N, T))
lines.append('# planner total %.1f us\n' % (sum(q['ms'] for q in base) * 1e3))
best_plan = []
for layer in (1, 2, 3, 4, 5, 10, 11, 12, 13, 14):
    t0, desc0, _ = layer_us('', layer)
    lines.append('%-11s planner            %8.2f  %s\n' % (NAMES[layer], t0, desc0))
    best = (t0, 'planner')
    for tile in ((5,) if cout[layer] % 128 else (1, 6, 3, 4)):
        for kg in (1, 2):
            for sp in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32):
                try:
                    t, desc, _ = layer_us('%d:%d:%d:%d' % (layer, tile, sp, kg), layer, 3)
                except Exception as e:                      # an illegal combination is refused by the library, not run
                    lines.append('%-11s %-8s kg%d s%-3d  refused: %s\n' % (NAMES[layer], TILES[tile], kg, sp, str(e)[:60]))
                    continue
                lines.append('%-11s %-8s kg%d s%-3d %8.2f  %s\n' % (NAMES[layer], TILES[tile], kg, sp, t, desc))
                if t < best[0]:
                    best = (t, '%d:%d:%d:%d' % (layer, tile, sp, kg))
                if sp >= 4 and t > 2.0 * t0:
                    break                                   # more splits only get worse from here
    lines.append('%-11s BEST %s %.2f us (planner %.2f)\n' % (NAMES[layer], best[1], best[0], t0))
    if best[1] != 'planner' and best[0] < 0.97 * t0:
        best_plan.append(best[1])
os.environ['RY_PLAN'] = ','.join(best_plan)
net.set_dtype('bf16x3')
st = net.profile(1, T, 6)
lines.append('# RY_PLAN=%s total %.1f us (planner %.1f us)\n' % (os.environ['RY_PLAN'], sum(q['ms'] for q in st) * 1e3, sum(q['ms'] for q in base) * 1e3))
Path(OUT).parent.mkdir(parents=True, exist_ok=True)
open(OUT, 'w').writelines(lines)
sys.stdout.writelines(l for l in lines if 'BEST' in l or l.startswith('#'))
net.close()
