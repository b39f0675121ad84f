"""Consumer of `tests/golden/upstream/` -- golden vectors written by `scripts/make_upstream_goldens.py` from the REAL `yukarin` /
`become_yukarin` / `pysptk` (the arithmetic behind /root/reference/realtime_voice_conversion/yukarin_wrapper/voice_changer.py:24-42).

* directory absent (the state of this repository: the packages cannot be installed where it is built) -> every test here SKIPS and says
  so: parity stays UNPINNED;
* directory present with `provider: real` -> the CPU suite holds the ORACLE to the goldens (is the restatement right?), `-m gpu` holds the
  HIP path to them through the shims (the pin proper), both at the north-star bar of 1e-4;
* a manifest written by `--provider shim` is a dry run of the plumbing and is refused as a pin.

`test_generator_and_consumer_dry_run_emu` runs the generator against this repository's own shims on the emulator into a temporary
directory and pushes the result through the very same consumer functions, so that the door is known to open before somebody with the
real packages walks through it."""
import importlib.util
import json
from pathlib import Path

import numpy
import pytest

from oracle import effective_frame as oef
from oracle import mc2sp as omc
from oracle import torch_ref
from realtime_yukarin_amd import engine, synth
from realtime_yukarin_amd.weights import flatten_params

ROOT = Path(__file__).resolve().parent.parent
UPSTREAM = ROOT / 'tests' / 'golden' / 'upstream'
TOL = 1e-4
SKIP = ('tests/golden/upstream/ is absent: PARITY UNPINNED.  Run scripts/make_upstream_goldens.py on a machine that has the real yukarin / '
        'become-yukarin / pysptk and commit its output')


def load_generator():
    spec = importlib.util.spec_from_file_location('make_upstream_goldens', str(ROOT / 'scripts' / 'make_upstream_goldens.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def manifest_of(d: Path, allow_dry_run=False):
    if not (d / 'MANIFEST.json').exists():
        pytest.skip(SKIP)
    m = json.loads((d / 'MANIFEST.json').read_text())
    if m['provider'] != 'real' and not allow_dry_run:
        pytest.fail('tests/golden/upstream/ was written by a shim dry run (%s): that is not a pin, remove it' % m['provider'])
    gen = load_generator()
    for name, sums in m['models'].items():                    # the goldens belong to the weights this repository generates from the same seeds
        (d1, P1), (d2, P2) = synth.model_params(name)
        assert gen.sha(flatten_params(d1, P1)) == sums['stage1'] and gen.sha(flatten_params(d2, P2)) == sums['stage2'], name
    return m


def model_cases(m, d: Path, names):
    for case in m['cases']:
        name = {'syn8': 'SYN-8', 'syn64': 'SYN-64'}.get(case.split('_')[0])
        if name in names:
            yield name, int(case.split('_n')[1]), dict(numpy.load(str(d / (case + '.npz'))))


# ------------------------------------------------------------------ the oracle against the goldens (CPU)
def check_oracle(d: Path, m, names, tol=TOL):
    n_checked = 0
    for fs in (16000, 24000):
        g = dict(numpy.load(str(d / ('mc2sp_fs%d.npz' % fs))))
        assert round(float(g['alpha']), 3) == round(omc.mcepalpha(fs), 3)
        assert float(numpy.abs(omc.mc2sp_sptk(g['mc'], float(g['alpha']), int(g['fftlen'])) / g['sp'] - 1).max()) < 1e-9
        assert float(numpy.abs(omc.mc2sp_closed(g['mc'], float(g['alpha']), int(g['fftlen'])) / g['sp'] - 1).max()) < 1e-9
    for case in (c for c in m['cases'] if c.startswith('gate_')):
        g = dict(numpy.load(str(d / (case + '.npz'))))
        for thr in (40, 60, 80):
            want = g['thr%d' % thr]
            got = oef.separate_effective_mask(g['wave'], int(g['fs']), len(want), thr, 1024, m['frame_period'])
            assert numpy.array_equal(got, want), (case, thr)
    nets = {}
    for name, n, g in model_cases(m, d, names):
        if name not in nets:
            (_, P1), (_, P2) = synth.model_params(name)
            nets[name] = torch_ref.TorchUNet(P1), torch_ref.TorchUNet(P2)
        t1, t2 = nets[name]
        e1 = float(numpy.abs(torch_ref.stage1_convert_core(t1, g['mc']) - g['stage1_mc']).max() / numpy.abs(g['stage1_mc']).max())
        e2 = float(numpy.abs(torch_ref.stage2_convert(t2, g['stage2_in']) / g['stage2_out'].astype(numpy.float64) - 1).max())
        eff = oef.separate_effective_mask(g['wave'], m['fs_in'], n, float(g['threshold']), 1024, m['frame_period'])
        assert numpy.array_equal(eff, g['vc_effective'])
        mc = numpy.zeros((n, synth.MC_DIMS), numpy.float32)
        if eff.any():
            mc[eff] = torch_ref.stage1_convert_core(t1, g['mc'][eff])
        mid = omc.mc2sp(mc, omc.mcepalpha(m['fs_out']), 1024)
        e_mid = float(numpy.abs(mid / g['vc_mid_sp'] - 1).max())
        sp = torch_ref.stage2_convert(t2, (mid + 1e-16).astype(numpy.float32))
        e3 = float(numpy.abs(sp / g['vc_sp'].astype(numpy.float64) - 1).max())
        print('oracle vs upstream, %s n=%d: stage-1 %.2e, stage-2 %.2e, mc2sp of the window %.2e, whole window %.2e' % (name, n, e1, e2, e_mid, e3))
        assert max(e1, e2, e_mid, e3) < tol
        if 'ref_vc_sp' in g:                                     # the reference's own class must agree with its steps written out
            assert numpy.array_equal(g['ref_vc_sp'], g['vc_sp']) and numpy.array_equal(g['ref_vc_mc'], g['vc_mc'])
        n_checked += 1
    return n_checked


# ------------------------------------------------------------------ the HIP path (through the shims) against the goldens
def check_shims(d: Path, m, names, tmp_path, tol=TOL):
    from realtime_yukarin_amd import compat
    compat.install()
    from realtime_yukarin_amd.voice_changer import VoiceChanger
    gen = load_generator()
    n_checked = 0
    for name in names:
        cases = list(model_cases(m, d, [name]))
        if not cases:
            continue
        md = tmp_path / name
        md.mkdir()
        gen.write_models(md, name, m['fs_in'], m['fs_out'])
        prov = gen.Provider('shim')
        ac, sr = prov.converters(md, m['fs_out'], 0)
        vc = VoiceChanger(acoustic_converter=ac, super_resolution=sr, threshold=60)
        for _, n, g in cases:
            feat = dict(f0=g['f0'], ap=g['ap'], mc=g['mc'], voiced=g['voiced'])
            y1 = ac.convert(prov.feature(g['wave'], feat, m['fs_in']))
            e1 = float(numpy.abs(y1.mc - g['stage1_mc']).max() / numpy.abs(g['stage1_mc']).max())
            e2 = float(numpy.abs(sr.convert(g['stage2_in']).astype(numpy.float64) / g['stage2_out'] - 1).max())
            out = vc.convert_from_acoustic_feature(prov.feature(g['wave'], feat, m['fs_in']))
            e3 = float(numpy.abs(out.sp.astype(numpy.float64) / g['vc_sp'] - 1).max())
            e4 = float(numpy.abs(out.mc - g['vc_mc']).max() / max(float(numpy.abs(g['vc_mc']).max()), 1e-30))
            print('HIP path vs upstream, %s n=%d: stage-1 %.2e, stage-2 %.2e, whole window sp %.2e / mc %.2e' % (name, n, e1, e2, e3, e4))
            assert max(e1, e2, e3, e4) < tol
            assert numpy.allclose(out.f0, g['vc_f0'], rtol=1e-6) and numpy.array_equal(out.ap, g['vc_ap'])
            n_checked += 1
        vc.close(); ac.close(); sr.close()
    return n_checked


def test_oracle_against_the_upstream_goldens():
    m = manifest_of(UPSTREAM)
    assert check_oracle(UPSTREAM, m, ['SYN-8', 'SYN-64']) > 0


@pytest.mark.gpu
def test_hip_path_against_the_upstream_goldens_gpu(gpu_ctx, tmp_path):
    m = manifest_of(UPSTREAM)
    assert check_shims(UPSTREAM, m, ['SYN-8', 'SYN-64'], tmp_path) > 0


def test_a_dry_run_is_not_accepted_as_a_pin(tmp_path):
    (tmp_path / 'MANIFEST.json').write_text(json.dumps(dict(provider='shim (DRY RUN of the plumbing: NOT a pin)', models={}, cases=[])))
    with pytest.raises(pytest.fail.Exception):
        manifest_of(tmp_path)
    gen = load_generator()
    with pytest.raises(SystemExit):
        gen.main(['--provider', 'shim'])                         # refuses to write a dry run into tests/golden/upstream/


def test_generator_and_consumer_dry_run_emu(tmp_path, emu_ctx, monkeypatch):
    """generator (--provider shim, emulator) -> manifest + npz -> the consumer's oracle leg and shim leg: the plumbing works end to end."""
    monkeypatch.setattr(engine, 'get_context', lambda device=0, lib=None: emu_ctx)
    gen = load_generator()
    monkeypatch.setitem(gen.CASES, 'SYN-8', [40])
    out = tmp_path / 'dry'
    m = gen.main(['--provider', 'shim', '--out', str(out), '--models', 'SYN-8', '--gpu', '0'])
    assert m['provider'].startswith('shim') and (out / 'MANIFEST.json').exists() and len(m['cases']) == 2 + 2 + 1
    m = manifest_of(out, allow_dry_run=True)
    assert check_oracle(out, m, ['SYN-8']) == 1
    (tmp_path / 'models').mkdir()
    assert check_shims(out, m, ['SYN-8'], tmp_path / 'models', tol=1e-5) == 1
