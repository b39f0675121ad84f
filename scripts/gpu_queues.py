"""Which pairs of fresh HIP streams run side by side on this box?  usage: python scripts/gpu_queues.py [n_streams] [torch]"""
import os, sys
from pathlib import Path
import numpy
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
if 'torch' in sys.argv:
    torch.cuda.set_device(0); torch.cuda.synchronize(); _t = torch.zeros(8, device='cuda')
from realtime_yukarin_amd import engine, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = engine.get_context(0)
r = numpy.zeros((n, n), numpy.float32)
for rep in range(2):
    ctx.lib.check(ctx.lib.dll.ry_debug_stream_overlap(ctx.handle, n, 300, _lib._fptr(r)))
    print('GPU_MAX_HW_QUEUES=%s torch=%s rep %d' % (os.environ.get('GPU_MAX_HW_QUEUES'), 'torch' in sys.argv, rep))
    for i in range(n):
        print(' '.join('%4.2f' % v for v in r[i]))
