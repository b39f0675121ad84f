"""Host logic around the C ABI (no device work): the K-list weight validator refuses anything that is not exactly the
predictor's parameter set, the flat blob round-trips, and the size / FLOP bookkeeping reproduces SURVEY.md section 8's numbers
(the figures bench.py's roofline is computed from)."""
import numpy
import pytest

from realtime_yukarin_amd import netspec, sptk, synth, weights
from realtime_yukarin_amd.netspec import NetDesc


def small():
    d = NetDesc(1, 9, 9, 8, 8)
    return d, weights.synthetic_params(d, 3, bias_std=0.05)


def test_validator_accepts_the_klist_and_ignores_the_bn_counter():
    d, P = small()
    P['encoder/c3/batchnorm/N'] = numpy.array(12)                 # Chainer's save_npz also stores it
    weights.validate_params(d, P)
    assert weights.flatten_params(d, P).size == netspec.param_count(d)


@pytest.mark.parametrize('damage', ['missing', 'extra', 'shape', 'swapped'])
def test_validator_refuses_a_mismatch(damage, tmp_path):
    d, P = small()
    if damage == 'missing':
        del P['decoder/c4/batchnorm/avg_var']
    elif damage == 'extra':
        P['encoder/c8/c/W'] = numpy.zeros((1, 1, 4), 'f4')
    elif damage == 'shape':
        P['encoder/c2/c/W'] = P['encoder/c2/c/W'][:, :, :3]
    else:                                                         # a stage-1 file offered to a stage-2 predictor
        d = NetDesc(2, 1, 1, 8, 8)
    with pytest.raises(ValueError):
        weights.validate_params(d, P)
    weights.save_npz(tmp_path / 'm.npz', P)
    with pytest.raises(ValueError):
        weights.load_npz(d, tmp_path / 'm.npz')


def test_blob_round_trip_and_npz_round_trip(tmp_path):
    d, P = small()
    blob = weights.flatten_params(d, P)
    Q = weights.unflatten_params(d, blob)
    assert list(Q) == [k for k, _ in netspec.param_list(d)] and all(numpy.array_equal(Q[k], P[k]) for k in Q)
    with pytest.raises(ValueError):
        weights.unflatten_params(d, blob[:-1])
    weights.save_npz(tmp_path / 'm.npz', {k: v.astype('f8') for k, v in P.items()})          # a float64 file is cast, not refused
    L = weights.load_npz(d, tmp_path / 'm.npz')
    assert all(L[k].dtype == numpy.float32 and numpy.array_equal(L[k], P[k]) for k in L)


def test_parameter_counts_and_flops_match_the_survey():
    d1, d2 = synth.model_descs('SYN-64')
    assert round(netspec.param_count(d1) / 1e6, 1) == 13.6 and round(netspec.param_count(d2) / 1e6, 1) == 54.4
    assert round(netspec.flops(d1, 1024) / 1e9, 3) == 1.453 and round(netspec.flops(d1, 384) / 1e9, 3) == 0.545
    assert round(netspec.flops(d2, 128, 512) / 1e9, 1) == 47.4 and round(netspec.flops(d2, 384, 512) / 1e9, 1) == 142.2
    assert round(netspec.flops(d2, 512, 512) / 1e9, 1) == 189.6 and round(netspec.flops(d2, 128, 512) / 128 / 1e6) == 370   # MFLOP per padded frame
    d8 = NetDesc(2, 1, 1, 8, 8)
    assert round(netspec.param_count(d8) / 1e6, 2) == 0.85


def test_pad_rule_and_extensive_layers():
    assert [netspec.pad_frames(n) for n in (1, 100, 127, 128, 300, 1000)] == [127, 28, 1, 128, 84, 24]
    d = NetDesc(1, 9, 9, 8, 3)
    assert [netspec.enc_sample(d, i) for i in (1, 2, 3, 7)] == ['down', 'down', 'same', 'same']
    assert [netspec.dec_sample(d, j) for j in (0, 4, 5, 6)] == ['same', 'same', 'up', 'up']


def test_mcepalpha_reproduces_the_known_table():
    assert [round(sptk.mcepalpha(fs), 3) for fs in (16000, 24000, 44100)] == [0.41, 0.466, 0.544]
    m = sptk.mc2sp_matrix(8, 0.41, 1024)
    assert m.shape == (9, 513) and m.dtype == numpy.float64
    mc = numpy.random.default_rng(0).normal(size=(5, 9)) * 0.3
    assert numpy.allclose(numpy.exp(mc @ m), sptk.mc2sp(mc, 0.41, 1024), rtol=1e-10)


def test_rendezvous_name_is_the_launchers_identity_not_an_mtime(tmp_path, monkeypatch):
    """Round-3 advisor: the RCCL id file of `dist.NativeComm` was accepted by its mtime against the mtime of /proc/<ppid>, which procfs
    stamps at first lookup.  Now the NAME carries the launcher's (pid, start time) and a serial number, and nothing is compared by time:
    the identity is stable within a launch, differs between two launchers, and sits in a 0700 directory of this user."""
    import os
    import subprocess
    import sys
    from realtime_yukarin_amd import dist as rdist
    monkeypatch.delenv('RY_COMM_RENDEZVOUS', raising=False)
    monkeypatch.delenv('RY_COMM_NONCE', raising=False)
    monkeypatch.setenv('TMPDIR', str(tmp_path))
    import tempfile
    monkeypatch.setattr(tempfile, 'tempdir', None)
    a, b = rdist._launcher_identity(), rdist._launcher_identity()
    ppid, start = a.split('-')
    assert a == b and int(ppid) == os.getppid() and int(start) > 0
    code = 'import sys; sys.path.insert(0, %r); from realtime_yukarin_amd import dist; print(dist._launcher_identity())' % str(rdist.__file__).rsplit('/', 2)[0]
    child = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120).stdout.strip()
    assert child.split('-')[0] == str(os.getpid()) and child != a                  # another launcher, another name
    p = rdist._rendezvous_path()
    d = os.path.dirname(p)
    assert os.path.basename(p).startswith('comm_' + a + '_') and (os.stat(d).st_mode & 0o077) == 0 and d.startswith(str(tmp_path))
    monkeypatch.setenv('RY_COMM_NONCE', 'launch-42')
    assert 'comm_launch-42_' in rdist._rendezvous_path()


def test_rendezvous_record_of_a_dead_writer_is_not_an_id(monkeypatch):
    """Round-4 advisor: one launcher may start its ranks twice (torchrun --max-restarts, a test process); the file a crashed attempt left under
    the launch's name carries a dead RCCL id.  The record now names its writer (pid + start time) and counts only while that very process is
    alive; the elastic restart count is part of the name."""
    import os
    import subprocess
    import sys
    from realtime_yukarin_amd import dist as rdist
    rec = rdist._id_record(b'\x07' * 128)
    assert len(rec) == 160 and rec[:128] == b'\x07' * 128 and rdist._id_record_is_live(rec)
    child = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(60)'])
    try:
        start = rdist._proc_start(child.pid)
        other = b'\x07' * 128 + ('%d %s' % (child.pid, start)).encode().ljust(32, b' ')
        assert rdist._id_record_is_live(other)                                       # a live writer
        wrong = b'\x07' * 128 + ('%d %s' % (child.pid, str(int(start) + 1))).encode().ljust(32, b' ')
        assert not rdist._id_record_is_live(wrong)                                   # the pid was recycled: another start time
    finally:
        child.kill(); child.wait()
    assert not rdist._id_record_is_live(other)                                       # the writer is gone: a leftover
    assert not rdist._id_record_is_live(b'\x07' * 128) and not rdist._id_record_is_live(b'\x07' * 128 + b'garbage'.ljust(32, b' '))
    # round-5 advisor: where the check cannot be made (the writer could not read /proc and wrote start '0', or this reader cannot read its own entry)
    # the record is accepted, as before the check existed -- otherwise every other rank would spin for the full timeout
    assert rdist._id_record_is_live(b'\x07' * 128 + b'999999 0'.ljust(32, b' '))
    monkeypatch.setattr(rdist, '_proc_start', lambda pid: None)
    assert rdist._id_record_is_live(other)
    monkeypatch.undo()
    monkeypatch.delenv('RY_COMM_RENDEZVOUS', raising=False)
    monkeypatch.setenv('TORCHELASTIC_RESTART_COUNT', '0'); a = rdist._rendezvous_path()
    monkeypatch.setenv('TORCHELASTIC_RESTART_COUNT', '1'); b = rdist._rendezvous_path()
    assert a != b


def test_bench_line_stays_short_enough_for_the_drivers_tail():
    """The driver keeps an 8 KB tail of bench.py's output; round 3's 14 KB line lost `device_ms_per_step_rank0` that way.  The printed line
    is built by `bench.compact_line` from the long form: with every optional block present (taken from the committed round-4 details file) it
    must stay under 6 KB and keep the headline, every bracket and the two required objects."""
    import json
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    import bench
    full = json.loads((root / 'profiles' / 'r04/d_bench_details.json').read_text())
    line = bench.compact_line(full, 'gpurun_out/bench_details_1gpu.json')
    text = json.dumps(line)
    assert len(text) < 6144, len(text)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
              'config', 'roofline', 'cpu_baseline', 'brackets', 'spread', 'slow_brackets', 'device_ms_per_step_rank0', 'small_window'):
        assert k in line, k
    assert len(line['brackets']) == line['repeats'] == 7 and all(set(b) == {'wall_ms', 'enq_ms', 'dev_ms'} for b in line['brackets'])
    assert set(('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')) <= set(line['roofline'])
    assert set(('value', 'unit', 'cores', 'kind', 'sample')) <= set(line['cpu_baseline'])
    assert 'kernels' in full and 'kernels' not in line and 'note' not in json.dumps(line['small_window'])
