#!/usr/bin/env python3
"""Generates tests/golden/gate/*.npz: seeded waves + the ORACLE's silence masks (`oracle/effective_frame.py`, the loop-per-frame
restatement of `separate_effective`, /root/reference/realtime_voice_conversion/yukarin_wrapper/voice_changer.py:27-31).

PARITY UNPINNED like every fixture here: librosa / yukarin cannot be imported in this container, so the masks come from our own
restatement ([MEM]: frame power = librosa.feature.rms(center=True, pad 'reflect') ** 2 in the wave's dtype, numpy's pairwise sum;
gate = power_to_db(ref 1.0, top_db 80) > -threshold; 'max' = the relative form).  They pin the oracle against regressions and give the
shim, the emulator and the GPU a committed target that does not depend on the oracle code at test time.
Run from the repo root: `python tests/golden/make_gate_golden.py`."""
import sys
from pathlib import Path

import numpy

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))

from oracle import effective_frame as oef  # noqa: E402

OUT = Path(__file__).resolve().parent / 'gate'
FS, FP, HOP = 16000, 5, 80


def main():
    OUT.mkdir(exist_ok=True)
    rng = numpy.random.default_rng(20260926)
    w = (0.1 * rng.normal(size=120 * HOP)).astype(numpy.float32)
    w[20 * HOP:50 * HOP] *= 1e-4
    w[80 * HOP:100 * HOP] = 0.0
    ramp = (numpy.geomspace(1e-6, 0.5, 60 * HOP) * rng.normal(size=60 * HOP)).astype(numpy.float32)
    cases = {'speech120': w, 'ramp60': ramp, 'short300': w[:300].copy(), 'tiny37': w[:37].copy(), 'ragged1234': w[:1234].copy()}
    for name, wave in cases.items():
        n = len(wave) // HOP + 1
        masks = {}
        for ref in ('abs', 'max'):
            for thr in (20, 60, 80, 100):
                for fft in (1024, 256):
                    masks['%s_thr%d_fft%d' % (ref, thr, fft)] = oef.separate_effective_mask(wave, FS, n, thr, fft, FP, ref)
        numpy.savez_compressed(OUT / ('%s.npz' % name), wave=wave, n_frames=n, **masks)
    print('wrote', sorted(p.name for p in OUT.glob('*.npz')))


if __name__ == '__main__':
    main()
