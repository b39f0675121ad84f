"""Drop-in check: the reference's own model-free unittest files run UNCHANGED on top of the `yukarin` /
`become_yukarin` / `chainer` shims (five of the six test files need no trained model).
Only possible where /root/reference is mounted (this container); skipped on the GPU box."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REF = Path('/root/reference')

FILES = ['test_segment', 'test_base_stream', 'test_encode_stream', 'test_convert_stream', 'test_feature_wrapper_segment_method']


@pytest.mark.skipif(not REF.exists(), reason='/root/reference is not mounted here')
@pytest.mark.parametrize('name', FILES)
def test_reference_unittest_file_passes_on_shims(name):
    env = dict(os.environ)
    env['PYTHONPATH'] = os.pathsep.join([str(ROOT / 'realtime_yukarin_amd' / 'compat'), str(ROOT), str(ROOT / 'tests' / 'stubs'), str(REF)])
    r = subprocess.run([sys.executable, '-m', 'unittest', '-v', 'tests.%s' % name], cwd=str(REF), env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert 'OK' in r.stdout


@pytest.mark.skipif(not REF.exists(), reason='/root/reference is not mounted here')
def test_reference_hot_path_modules_import_on_shims():
    """convert_worker / yukarin_converter / check.py import cleanly (chainer stub, config loaders, F0Converter)."""
    env = dict(os.environ)
    env['PYTHONPATH'] = os.pathsep.join([str(ROOT / 'realtime_yukarin_amd' / 'compat'), str(ROOT), str(ROOT / 'tests' / 'stubs'), str(REF)])
    code = ('import realtime_voice_conversion.worker.convert_worker as w; '
            'import realtime_voice_conversion.converter.yukarin_converter as c; '
            'import realtime_voice_conversion.yukarin_wrapper.voice_changer as v; import check; print("imports ok")')
    r = subprocess.run([sys.executable, '-c', code], cwd=str(REF), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       universal_newlines=True, timeout=300)
    assert r.returncode == 0 and 'imports ok' in r.stdout, r.stdout[-3000:]
