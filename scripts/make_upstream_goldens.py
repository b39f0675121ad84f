#!/usr/bin/env python3
"""Writes golden vectors of the convert hot path from the REAL upstream packages -- the door to a pinned parity.

Why: the arithmetic of the path (`VoiceChanger.convert_from_acoustic_feature`, /root/reference/realtime_voice_conversion/yukarin_wrapper/
voice_changer.py:24-42) lives in `yukarin`, `become-yukarin` (un-pinned git dependencies, /root/reference/requirements.txt:7-8), `chainer`
and `pysptk`, none of which can be installed where this repository is built.  Everything under `oracle/` is therefore a restatement and
every parity figure says "unpinned".  On a machine that HAS those packages this script runs them -- the real `AcousticConverter.convert`,
`SuperResolution.convert`, `pysptk.mc2sp` / `pysptk.util.mcepalpha`, `Wave.get_effective_frame`, and (with --reference-root) the
reference's own unchanged `VoiceChanger` -- on the seeded synthetic models and inputs this repository's tests use, and writes

    tests/golden/upstream/MANIFEST.json                 provider, package versions, seeds, weight checksums, the case list
    tests/golden/upstream/<case>.npz                    inputs + the upstream outputs

`tests/test_upstream_goldens.py` consumes them when present: CPU suite = the oracle against the goldens (is the restatement right?),
`-m gpu` = the HIP path against the goldens (the pin proper).  Until somebody commits that directory the test skips with a loud reason.

    python scripts/make_upstream_goldens.py                         # real packages, CPU (gpu=None), writes tests/golden/upstream/
    python scripts/make_upstream_goldens.py --reference-root /path/to/realtime-yukarin      # + the reference's VoiceChanger end to end
    python scripts/make_upstream_goldens.py --provider shim --out /tmp/dry --models SYN-8   # DRY RUN of the plumbing against this
                                                                                            # repository's own shims (NOT a pin;
                                                                                            # the manifest says so and the consumer
                                                                                            # refuses to count it as one)

Model files are written in the layout the upstream constructors read ([MEM]): `chainer.serializers.save_npz` of the bare predictor (the
K-list of SURVEY.md section 8(c) item 3) + a config.json per stage.  The real `create_from_json` may require keys beyond the ones the
reference reads; CONFIG_EXTRA below carries the training-side keys as far as they are known -- if the installed version wants more, add
them there (they do not influence inference) and note it in the manifest (`--note`)."""
import argparse
import hashlib
import importlib
import json
import sys
import tempfile
import time
from pathlib import Path

import numpy

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from realtime_yukarin_amd import synth                         # noqa: E402  (seeds, shapes, the canonical model configs: data, not arithmetic)
from realtime_yukarin_amd.weights import flatten_params, save_npz   # noqa: E402

FRAME_PERIOD, ORDER = 5, 8
CASES = {                                                       # model -> windows (real frames); BASELINE.json configs #3/#4, #5, #1, N % 128 == 0
    'SYN-8': [60, 100, 128, 300],
    'SYN-64': [100, 128, 300, 400, 600],
}
# training-side keys of the upstream config schemas ([MEM]); inference ignores them
CONFIG_EXTRA = {
    'stage1_dataset': dict(input_glob='', target_glob='', indexes_glob='', train_crop_size=512, input_global_noise=0.0, input_local_noise=0.0,
                           target_global_noise=0.0, target_local_noise=0.0, seed=0, num_test=1),
    'stage1_model': dict(discriminator_base_channels=32, discriminator_extensive_layers=5, weak_discriminator=False, glu_generator=False),
    'stage1_rest': dict(loss=dict(mse=100, adversarial=1), train=dict(batchsize=8, gpu=-1, log_iteration=100, snapshot_iteration=1000, stop_iteration=None,
                        optimizer=dict(alpha=0.0002, beta1=0.5, beta2=0.999, name='adam')), project=dict(name='', tags=[])),
    'stage2_dataset': dict(input_glob='', train_crop_size=512, input_global_noise=0.0, input_local_noise=0.0, blur_size_factor=0, seed=0, num_test=1),
    'stage2_model': dict(discriminator_base_channels=32, discriminator_extensive_layers=5),
    'stage2_rest': dict(loss=dict(mse=100, adversarial=1), train=dict(batchsize=8, gpu=-1, log_iteration=100, snapshot_iteration=1000), project=dict(name='', tags=[])),
}


def sha(a: numpy.ndarray) -> str:
    return hashlib.sha256(numpy.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def write_models(d: Path, name: str, fs_in: int, fs_out: int):
    """SYN weights (seeds 356 / 357) as upstream-style model files; returns the checksums of the flat blobs."""
    (d1, P1), (d2, P2) = synth.model_params(name)
    save_npz(d / 'stage1.npz', P1)
    save_npz(d / 'stage2.npz', P2)
    acoustic_param = dict(sampling_rate=fs_in, pad_second=0, threshold_db=None, frame_period=FRAME_PERIOD, order=ORDER,
                          alpha={16000: 0.41, 24000: 0.466}[fs_in], f0_floor=71, f0_ceil=800, fft_length=1024, dtype='float32')
    c1 = dict(dataset=dict(acoustic_param=acoustic_param, in_features=['mc'], out_features=['mc'], **CONFIG_EXTRA['stage1_dataset']),
              model=dict(in_channels=d1.in_ch, out_channels=d1.out_ch, generator_base_channels=d1.base,
                         generator_extensive_layers=d1.extensive_layers, **CONFIG_EXTRA['stage1_model']), **CONFIG_EXTRA['stage1_rest'])
    c2 = dict(dataset=dict(param=dict(voice_param=dict(sample_rate=fs_out, top_db=None, pad_second=0.0),
                                      acoustic_feature_param=dict(frame_period=FRAME_PERIOD, order=ORDER, alpha={16000: 0.41, 24000: 0.466}[fs_out],
                                                                  f0_estimating_method='harvest')), **CONFIG_EXTRA['stage2_dataset']),
              model=dict(generator_base_channels=d2.base, generator_extensive_layers=d2.extensive_layers, **CONFIG_EXTRA['stage2_model']),
              **CONFIG_EXTRA['stage2_rest'])
    (d / 'stage1.json').write_text(json.dumps(c1))
    (d / 'stage2.json').write_text(json.dumps(c2))
    numpy.save(str(d / 'f0_in.npy'), {'mean': numpy.log(200.0), 'var': 0.04})
    numpy.save(str(d / 'f0_out.npy'), {'mean': numpy.log(300.0), 'var': 0.09})
    return dict(stage1=sha(flatten_params(d1, P1)), stage2=sha(flatten_params(d2, P2))), (c1, c2)


def window(n: int, seed: int, fs: int):
    """A window as the encode stage hands it over: raw wave (loud / silent / below-the-gate stretches) + features of every frame."""
    hop = fs * FRAME_PERIOD // 1000
    rng = numpy.random.default_rng(seed)
    wave = (0.1 * rng.normal(size=n * hop)).astype(numpy.float32)
    a, b = n // 6, n // 2
    wave[a * hop:b * hop] = 0.0
    wave[(n - n // 8) * hop:] *= 1e-5
    f0 = numpy.where(rng.random((n, 1)) < 0.3, 0.0, rng.lognormal(numpy.log(220.0), 0.2, (n, 1))).astype(numpy.float32)
    return wave, dict(f0=f0, ap=rng.uniform(0.001, 0.999, (n, synth.FFT_BINS)).astype(numpy.float32),
                      mc=(rng.normal(size=(n, synth.MC_DIMS)) * synth.MC_SCALE).astype(numpy.float32), voiced=f0 > 0)


class Provider(object):
    """The classes under `yukarin` / `become_yukarin` / `pysptk` as importable right now, plus where they came from."""

    def __init__(self, kind: str, reference_root=None):
        self.kind = kind
        if kind == 'shim':
            from realtime_yukarin_amd import compat
            compat.install()
        else:
            compat = str(ROOT / 'realtime_yukarin_amd' / 'compat')
            sys.path[:] = [p for p in sys.path if str(Path(p).resolve()) != compat]
        self.yukarin = importlib.import_module('yukarin')
        self.become = importlib.import_module('become_yukarin')
        self.y_config = importlib.import_module('yukarin.config')
        self.sr_config = importlib.import_module('become_yukarin.config.sr_config')
        self.f0c = importlib.import_module('yukarin.f0_converter')
        is_shim = 'realtime_yukarin_amd' in str(getattr(self.yukarin, '__file__', ''))
        if (kind == 'real') == is_shim:
            raise SystemExit('--provider %s, but `import yukarin` resolved to %s' % (kind, self.yukarin.__file__))
        if kind == 'real':
            self.pysptk = importlib.import_module('pysptk')
            # one frame per call: every pysptk version accepts a 1-D float64 frame
            self.mc2sp = lambda mc, alpha, fftlen: numpy.stack([self.pysptk.mc2sp(numpy.ascontiguousarray(r, dtype=numpy.float64), alpha=alpha, fftlen=fftlen) for r in mc])
            self.mcepalpha = self.pysptk.util.mcepalpha
        else:
            from realtime_yukarin_amd import sptk
            self.mc2sp, self.mcepalpha = sptk.mc2sp, sptk.mcepalpha
        self.voice_changer = None
        if reference_root is not None:
            sys.path.insert(0, str(reference_root))
            self.voice_changer = importlib.import_module('realtime_voice_conversion.yukarin_wrapper.voice_changer').VoiceChanger

    def versions(self):
        out = {}
        for name in ('yukarin', 'become_yukarin', 'chainer', 'pysptk', 'librosa', 'numpy'):
            try:
                m = importlib.import_module(name)
                out[name] = dict(version=str(getattr(m, '__version__', '?')), file=str(getattr(m, '__file__', '?')))
            except ImportError:
                out[name] = None
        return out

    def converters(self, d: Path, fs_out: int, gpu):
        f0c = self.f0c.F0Converter(input_statistics=d / 'f0_in.npy', target_statistics=d / 'f0_out.npy')
        ac = self.yukarin.AcousticConverter(self.y_config.create_from_json(d / 'stage1.json'), d / 'stage1.npz', gpu=gpu, f0_converter=f0c,
                                            out_sampling_rate=fs_out)
        sr = self.become.SuperResolution(self.sr_config.create_from_json(d / 'stage2.json'), d / 'stage2.npz', gpu=gpu)
        return ac, sr

    def feature(self, wave, feat, fs):
        f = self.yukarin.AcousticFeature(**{k: v.copy() for k, v in feat.items()})
        f.wave = self.yukarin.Wave(wave=wave, sampling_rate=fs)
        return f


def generate(out: Path, provider: Provider, models, fs_in: int, fs_out: int, gpu, note: str):
    out.mkdir(parents=True, exist_ok=True)
    manifest = dict(provider='real' if provider.kind == 'real' else 'shim (DRY RUN of the plumbing: NOT a pin)', created=time.strftime('%Y-%m-%d %H:%M:%S'),
                    versions=provider.versions(), fs_in=fs_in, fs_out=fs_out, frame_period=FRAME_PERIOD, seeds=dict(stage1=synth.SEED_STAGE1, stage2=synth.SEED_STAGE2),
                    note=note, models={}, cases=[])

    def save(name, **arrays):
        numpy.savez_compressed(out / (name + '.npz'), **arrays)
        manifest['cases'].append(name)

    # ---- pysptk alone: the all-pass constants and mc2sp at both rates the reference uses
    rng = numpy.random.default_rng(synth.SEED_INPUT + 7)
    mc = (rng.normal(size=(64, synth.MC_DIMS)) * synth.MC_SCALE)
    for fs in (16000, 24000):
        alpha = float(provider.mcepalpha(fs))
        save('mc2sp_fs%d' % fs, mc=mc, alpha=numpy.float64(alpha), fftlen=numpy.int64(1024), sp=numpy.asarray(provider.mc2sp(mc, alpha, 1024), dtype=numpy.float64))
    # ---- the silence gate: Wave.get_effective_frame on waves with silent / quiet stretches, thresholds 40 / 60 / 80 dB
    for n, seed in ((120, 1), (300, 2)):
        wave, _ = window(n, seed, fs_in)
        masks = {('thr%d' % thr): numpy.asarray(provider.yukarin.Wave(wave=wave, sampling_rate=fs_in).get_effective_frame(
            threshold_db=thr, fft_length=1024, frame_period=FRAME_PERIOD), dtype=bool) for thr in (40, 60, 80)}
        save('gate_n%d' % n, wave=wave, fs=numpy.int64(fs_in), **masks)
    # ---- the two CNNs behind their convert() wrappers, and the whole window call
    for name in models:
        with tempfile.TemporaryDirectory() as tmp:
            d = Path(tmp)
            sums, _ = write_models(d, name, fs_in, fs_out)
            manifest['models'][name] = sums
            ac, sr = provider.converters(d, fs_out, gpu)
            for n in CASES[name]:
                wave, feat = window(n, 100 + n, fs_in)
                f_in = provider.feature(wave, feat, fs_in)
                y1 = ac.convert(f_in)                                               # stage-1: every frame, no gate
                sp_in = synth.stage2_input(n, seed=200 + n)[0]
                y2 = numpy.asarray(sr.convert(sp_in.copy()))                        # stage-2 on its own
                arrays = dict(wave=wave, f0=feat['f0'], ap=feat['ap'], mc=feat['mc'], voiced=feat['voiced'],
                              stage1_mc=numpy.asarray(y1.mc, dtype=numpy.float32), stage1_f0=numpy.asarray(y1.f0, dtype=numpy.float32),
                              stage2_in=sp_in, stage2_out=y2.astype(numpy.float32), threshold=numpy.float64(60))
                # voice_changer.py:24-42 step by step (and through the reference's own class when it is importable)
                f_eff, effective = ac.separate_effective(wave=f_in.wave, feature=provider.feature(wave, feat, fs_in), threshold=60)
                f_out = ac.convert(f_eff) if numpy.any(effective) else f_eff
                f_out = ac.combine_silent(effective=effective, feature=f_out)
                f_out = ac.decode_spectrogram(f_out)
                mid = numpy.array(numpy.asarray(f_out.sp), dtype=numpy.float64)
                f_out.sp += 1e-16
                sp = numpy.asarray(sr.convert(numpy.asarray(f_out.sp).astype(numpy.float32)))
                arrays.update(vc_effective=numpy.asarray(effective, dtype=bool), vc_mc=numpy.asarray(f_out.mc, dtype=numpy.float32),
                              vc_f0=numpy.asarray(f_out.f0, dtype=numpy.float32), vc_ap=numpy.asarray(f_out.ap, dtype=numpy.float32),
                              vc_mid_sp=mid, vc_sp=sp.astype(numpy.float32))
                if provider.voice_changer is not None:
                    vc = provider.voice_changer(acoustic_converter=ac, super_resolution=sr, threshold=60)
                    o = vc.convert_from_acoustic_feature(provider.feature(wave, feat, fs_in))
                    arrays.update(ref_vc_sp=numpy.asarray(o.sp, dtype=numpy.float32), ref_vc_mc=numpy.asarray(o.mc, dtype=numpy.float32))
                save('%s_n%d' % (name.lower().replace('-', ''), n), **arrays)
                print('%s n=%d: %d of %d frames effective' % (name, n, int(numpy.sum(effective)), n))
            for obj in (ac, sr):
                if hasattr(obj, 'close'):
                    obj.close()
    (out / 'MANIFEST.json').write_text(json.dumps(manifest, indent=1))
    print('wrote %d cases + MANIFEST.json to %s (provider: %s)' % (len(manifest['cases']), out, manifest['provider']))
    return manifest


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--out', type=Path, default=ROOT / 'tests' / 'golden' / 'upstream')
    ap.add_argument('--provider', choices=('real', 'shim'), default='real')
    ap.add_argument('--models', default='SYN-8,SYN-64')
    ap.add_argument('--fs-in', type=int, default=16000)
    ap.add_argument('--fs-out', type=int, default=24000, help='out_sampling_rate (converter/yukarin_converter.py:46 hard-codes 24000)')
    ap.add_argument('--gpu', type=int, default=None, help='device for the upstream constructors; default None = Chainer on the CPU (check.py:54-63)')
    ap.add_argument('--reference-root', type=Path, default=None, help='checkout of realtime-yukarin: also run its unchanged VoiceChanger')
    ap.add_argument('--note', default='')
    a = ap.parse_args(argv)
    if a.provider == 'shim' and a.out.resolve() == (ROOT / 'tests' / 'golden' / 'upstream').resolve():
        raise SystemExit('a shim dry run must not be written to tests/golden/upstream (it is not a pin): pass --out')
    provider = Provider(a.provider, a.reference_root)
    return generate(a.out, provider, [m for m in a.models.split(',') if m], a.fs_in, a.fs_out, a.gpu, a.note)


if __name__ == '__main__':
    main()
