#!/usr/bin/env python3
"""Per-layer times of the stage-1 predictor for a batch of windows (eager launches bracketed by HIP events): where a batched stage-1 call
spends its time.  usage (GPU box): python scripts/gpu_s1_batch_layers.py [batch] [frames]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from realtime_yukarin_amd import engine, synth                      # noqa: E402
from realtime_yukarin_amd.netspec import pad_frames                 # noqa: E402
from realtime_yukarin_amd.weights import flatten_params             # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
T = N + pad_frames(N)
(d1, P1), _ = synth.model_params('SYN-64')
ctx = engine.get_context(0)
net = engine.Net(ctx, d1, flatten_params(d1, P1))
for b in (1, B):
    st = net.profile(b, T, 5)
    print('# stage-1, batch %d, %d padded frames: sum of eager launches %.1f us' % (b, T, sum(q['ms'] for q in st) * 1e3))
    for q in st:
        print('%-12s %-36s grid=%-14s %8.2f us %8.2f TFLOP/s %9.1f GB/s' % (q['layer'], q['name'], 'x'.join(map(str, q['grid'])), q['ms'] * 1e3,
                                                                          q['flops'] / max(q['ms'], 1e-9) / 1e9, q['bytes'] / max(q['ms'], 1e-9) / 1e6))
net.close()
