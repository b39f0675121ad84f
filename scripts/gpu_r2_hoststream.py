"""Host-array stream through the pinned ring (ry_vc_submit / ry_vc_wait): ms per window by lanes and depth."""
import sys, time
from pathlib import Path
import numpy
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
torch.cuda.set_device(0)
from realtime_yukarin_amd import engine, sptk, synth
from realtime_yukarin_amd.weights import synthetic_params, flatten_params
N = 300
ctx = engine.get_context(0)
d1, d2 = synth.model_descs('SYN-64')
n1 = engine.Net(ctx, d1, flatten_params(d1, synthetic_params(d1, synth.SEED_STAGE1)))
n2 = engine.Net(ctx, d2, flatten_params(d2, synthetic_params(d2, synth.SEED_STAGE2)), width=synth.FFT_BINS - 1)
mtx = sptk.mc2sp_matrix(d1.out_ch - 1, sptk.mcepalpha(16000), 2 * (synth.FFT_BINS - 1))
xh = synth.stage1_input(N, 1)[0]; eff = numpy.ones(N, bool)
for lanes in (1, 2):
    core = engine.VcCore(n1, n2, mtx, lanes=lanes)
    for depth in (1, 2, 3, 6):
        for _ in core.convert_stream([(xh, eff)] * 12, depth=depth):
            pass
        res = []
        for rep in range(3):
            th = time.perf_counter()
            for _ in core.convert_stream([(xh, eff)] * 60, depth=depth):
                pass
            res.append((time.perf_counter() - th) / 60 * 1e3)
        print('lanes %d depth %d: %s ms per window' % (lanes, depth, ' '.join('%.4f' % r for r in res)), flush=True)
    core.close()
