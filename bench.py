#!/usr/bin/env python3
"""bench.py -- throughput of the convert hot path on N MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it under torch.distributed.run (one rank per
GPU, RCCL).  One "step" = one pass of the hot path over one window already resident in HBM, CHAINED as
`VoiceChanger.convert_from_acoustic_feature` chains it (/root/reference/realtime_voice_conversion/yukarin_wrapper/voice_changer.py:
33-41): effective frames -> stage-1 CNN -> combine_silent -> mc2sp (+1e-16) -> stage-2 CNN -> spectrogram in HBM
(`ry_vc_enqueue_device`).  Consecutive steps pipeline by themselves: the window call keeps up to six windows in flight over two
pairs of predictor streams (`ry_vc_set_lanes`, `--lanes`, default 2; `--lanes 1` = one stage-2 forward after the other), as consecutive
buffers of a live stream or the windows of run.py's queue would be.  Default window: 300 frames = buffer_time 0.5 s + 2 x
convert_extra_time 0.5 s at 5 ms frames (/root/reference/config.yaml:14; BASELINE config #3).  Windows are independent, so N GPUs take
N times the windows (weak scaling) with one RCCL broadcast of each weight blob at start-up and no collective in the timed region.
Prints ONE JSON line on rank 0.

`--force-dist` runs the N > 1 code path (process group, broadcast, barrier, all-reduce of the elapsed time) with ONE rank on one GPU;
`--comm native` takes RCCL through the C ABI (ry_comm_*) instead of torch.distributed; `--emulator` (tests only) runs the same script on
the host-side emulator build with gloo -- its numbers mean nothing.
"""
import argparse
import hashlib
import json
import os
import sys
import time
from pathlib import Path

import numpy

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 measured copy
F32_MFMA_PEAK_TF = 157.3       # v_mfma_f32_32x32x2_f32 dense peak (= fp32 vector peak)
SP_FLOOR = 1e-16               # voice_changer.py:39


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--repeats', type=int, default=7,
                    help='the K-step bracket is timed this many times back to back (same K, same fences); `value` comes from the MEDIAN bracket, every '
                         'bracket is listed in `brackets`')
    ap.add_argument('--details-out', default=None,
                    help='file for the long form of the result (per-kernel table, batch / host-path / discard measurements, notes); default '
                         'gpurun_out/bench_details_<N>gpu.json.  The printed line carries the headline, the brackets and the roofline objects only')
    ap.add_argument('--frames', type=int, default=300, help='real frames per window (N)')
    ap.add_argument('--extra-frames', type=int, default=None,
                    help='overlap frames on EACH side of the window that ConvertStream throws away (convert_stream.py:40-42); default 100 '
                         '(= convert_extra_time 0.5 s) for the 300-frame window, else 0')
    ap.add_argument('--windows', type=int, default=1, help='windows per GPU per step (converted one after the other)')
    ap.add_argument('--lanes', type=int, default=None, help='windows that run side by side on their own predictor streams (ry_vc_set_lanes; default RY_VC_LANES or 2)')
    ap.add_argument('--model', default=None)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16', 'bf16x3'],
                    help="stage-2 MFMA operand type ('bf16' = BASELINE config #5; 'bf16x3' = split-bf16, DESIGN.md 5.1); the headline is f32")
    ap.add_argument('--comm', default='torch', choices=['torch', 'native'], help='transport of the weight broadcast / barrier when N > 1')
    ap.add_argument('--force-dist', action='store_true', help='run the N > 1 code path with one rank (real RCCL init on one GPU)')
    ap.add_argument('--emulator', action='store_true', help='TESTS ONLY: emulator build of the kernels on the CPU, gloo; numbers are meaningless')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-split-bf16', action='store_true', help="skip the extra 'split_bf16' measurement of an f32 run")
    ap.add_argument('--no-extras', action='store_true', help='headline + roofline only (no host-path, cold-cache, batch or gate measurements)')
    ap.add_argument('--chained-batch', default='8', help='windows per call of the batched window call measured next to the headline (comma-separated)')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--profile-reps', type=int, default=5)
    ap.add_argument('--dispatcher', action='store_true',
                    help='measure the multi-GPU dispatcher instead (dispatch.ChunkDispatcher: host feature windows in, picked features out): null '
                         'workers G = 1..8 (the central loop alone), paced null workers, and on the GPU G = 1 and two worker processes on one GPU')
    ap.add_argument('--dispatcher-windows', type=int, default=400, help='windows per real-GPU dispatcher measurement')
    ap.add_argument('--dispatcher-depth', type=int, default=2, help='windows a worker keeps in flight on its GPU')
    ap.add_argument('--profile-only', action='store_true', help='only print the per-kernel-family profile of stage-2 (tuning aid)')
    ap.add_argument('--layers-out', default=None, help='write the per-launch profile (layer, kernel, ms, TFLOP/s, GB/s) to this file')
    return ap.parse_args(argv)


def source_hash():
    """Identifies the kernel build a PMC summary belongs to (profiles/*pmc_summary.txt carry it in their header)."""
    h = hashlib.sha1()
    for f in sorted((ROOT / 'realtime_yukarin_amd' / 'csrc').glob('ry_*')):          # every unit and header of libry355.so
        if f.suffix in ('.cpp', '.h'):
            h.update(f.read_bytes())
    return h.hexdigest()[:12]


def pmc_table():
    """{kernel name: (HBM bytes per launch, average us)} from the newest committed PMC summary made from THIS source (2 x FETCH_SIZE +
    WRITE_SIZE in KB per dispatch: gfx950 counts wide coalesced reads at half size, MI355X_MICROARCH.md).  Empty when no summary of
    the current kernels exists -- traffic is then reported as null rather than read from an older build."""
    import glob
    want = source_hash()
    for path in sorted(glob.glob(str(ROOT / 'profiles' / '**' / '*pmc_summary.txt'), recursive=True), reverse=True):
        lines = open(path).read().splitlines()
        if not any(('source ' + want) in ln for ln in lines[:6]):
            continue
        tab = {}
        for line in lines:
            if line.startswith('#') or line.startswith('kernel'):
                continue
            cols = line.split()
            try:
                tab[cols[0]] = ((2.0 * float(cols[-2]) + float(cols[-1])) * 1024.0, float(cols[2]))
            except (ValueError, IndexError):
                pass
        return tab, str(Path(path).relative_to(ROOT))
    return {}, None


def rocprof_table():
    """{kernel name: average microseconds per launch} from the newest committed `rocprofv3 --kernel-trace --stats` summary made from THIS
    kernel source, one window at a time (profiles/*kernel_stats.txt, not the two-lane trace): what `roofline.frac` is computed from, so that
    the figure in the line can be recomputed from profiles/.  Empty when no summary of the current source exists."""
    import glob
    want = source_hash()
    for path in sorted(glob.glob(str(ROOT / 'profiles' / '**' / '*kernel_stats.txt'), recursive=True), reverse=True):
        lines = open(path).read().splitlines()
        if not any(('source ' + want) in ln for ln in lines[:3]):
            continue
        tab = {}
        for line in lines:
            if line.startswith('#') or line.startswith('kernel'):
                continue
            cols = line.split()
            try:
                name = line[:line.index('(Ry')] if '(Ry' in line else line[:72]
                avg = float(cols[-2])
            except (ValueError, IndexError):
                continue
            tab[name.replace('void ', '').replace(' ', '')] = avg
        import re
        m = re.search(r'box (\S+) (\d{4}-\d\d-\d\d)', lines[0])          # where and when the summary was measured (scripts/gpu_profile.sh)
        return tab, str(Path(path).relative_to(ROOT)) + (' [measured on box %s, %s]' % m.groups() if m else '')
    return {}, None


def dispatcher_bench(args):
    """`--dispatcher`: the path a user of the reference runs on G GPUs -- `dispatch.ChunkDispatcher`, host feature windows in (what
    `ConvertStream.fetch` returns: wave + f0 / ap / mc / voiced of N frames), the picked features out (`ConvertStream.process` keeps the
    buffer in the middle, /root/reference/realtime_voice_conversion/stream/convert_stream.py:40-42; ordering contract /root/reference/run.py:171-183).
    windows/s of (a) the central loop alone: null workers that answer at once with a payload of the real size, G = 1..8; (b) the same with
    workers that take the GPU time of a window (paced: busy wait), i.e. what G GPUs could be fed; (c) one real worker on the GPU, (d) two
    worker processes on the one GPU, (e) the same windows through the in-process window call (no transport) for comparison."""
    import tempfile
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
    from realtime_yukarin_amd import dispatch, synth
    emu = args.emulator
    model = args.model or ('SYN-8' if emu else 'SYN-64')
    N = args.frames
    extra = args.extra_frames if args.extra_frames is not None else (100 if N == 300 else 0)
    pick = (extra, -extra, dispatch.PICK_KEYS) if extra > 0 else None
    tmp = Path(tempfile.mkdtemp(prefix='ry355-bench-'))
    synth.write_model_files(tmp, model)
    ac, sr = synth.build_converters(tmp)
    wins = [synth.feature_window(N, 900 + i, silent_stretch=(i % 3 == 2)) for i in range(9)]       # one window in three cut by the silence gate
    out = {'metric': 'dispatcher windows/s (host feature windows in, picked features out)', 'unit': 'windows/s', 'frames': N,
           'kept_frames': N - 2 * extra, 'model': model, 'host_cpus': os.cpu_count(), 'depth': args.dispatcher_depth,
           'bytes_up_per_window': int(sum(numpy.asarray(a).nbytes for a in (wins[0].wave.wave, wins[0].f0, wins[0].mc, wins[0].voiced))),
           'bytes_down_per_window': int((N - 2 * extra) * (synth.FFT_BINS * 4 + 4 + 1 + synth.MC_DIMS * 4) + N),
           'bytes_ap_not_shipped': int(wins[0].ap.nbytes + (N - 2 * extra) * synth.FFT_BINS * 4)}

    def rate(devices, n, **kw):
        with dispatch.ChunkDispatcher(ac, sr, devices, depth=args.dispatcher_depth, start_timeout=900, **kw) as d:
            for i in range(min(n, 40)):                                   # plans, graphs, rings warm
                d.submit(i, wins[i % 9], discard=(extra, extra), pick=pick); d.collect()
            d.drain(timeout=600)
            t0, got = time.perf_counter(), 0
            for i in range(n):
                d.submit(i, wins[i % 9], discard=(extra, extra), pick=pick)
                got += len(d.collect())
            got += len(d.drain(timeout=600))
            el = time.perf_counter() - t0
            assert got == n
            return n / el
    null_n = 200 if emu else 4000
    out['null_workers'] = {str(G): round(rate([0] * G, null_n, comm='host', null_workers=True), 1) for G in (1, 2, 4, 8)}
    out['null_workers_whole_objects'] = {str(G): round(rate([0] * G, null_n, comm='host', null_workers=True, lean=False), 1) for G in (1, 8)}
    gpu_ms = 0.88                                                         # one window on one MI355X with the discard hint
    if not emu and (os.cpu_count() or 1) >= 12:
        out['paced_null_workers'] = {'ms_per_window_per_worker': gpu_ms, 'windows_per_s': {
            str(G): round(rate([0] * G, 1000 * G, comm='host', null_workers=True, worker_hook=dispatch.paced(gpu_ms)), 1) for G in (1, 2, 4, 8)}}
    if not emu:
        nw = args.dispatcher_windows
        out['gpu_one_worker'] = round(rate([0], nw, comm='native'), 1)                # weights by the (one-rank) RCCL broadcast
        out['gpu_two_workers_one_gpu'] = round(rate([0, 0], nw, comm='host'), 1)
        # the same windows through the in-process window call, `depth` in flight (what one worker does, minus the rings)
        from realtime_yukarin_amd.voice_changer import VoiceChanger
        vc = VoiceChanger(acoustic_converter=ac, super_resolution=sr, threshold=60)
        import collections
        pend = collections.deque()

        def one_pass(n):
            for i in range(n):
                pend.append(vc.begin(wins[i % 9], discard=(extra, extra)))
                if len(pend) >= args.dispatcher_depth:
                    vc.finish(pend.popleft(), lean=True)
            while pend:
                vc.finish(pend.popleft(), lean=True)
        one_pass(40)
        t0 = time.perf_counter(); one_pass(nw); el = time.perf_counter() - t0
        out['in_process_window_call'] = round(nw / el, 1)
        out['gpu_one_worker_vs_in_process'] = round(out['gpu_one_worker'] / out['in_process_window_call'], 4)
        vc.close(); ac.close(); sr.close()
    out['null_ceiling_in_gpus'] = round(max(out['null_workers'].values()) / (1e3 / gpu_ms), 2)
    print(json.dumps(out), flush=True)
    return out


def self_launch(n, argv):
    """`python3 bench.py --gpus N` without a launcher: the same script under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free>` (one rank per GPU); the ranks' stdout / stderr are ours.  -> what rank 0 printed, parsed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), str(Path(__file__).resolve())] + argv
    print('bench.py: --gpus %d without a launcher: starting %s' % (n, ' '.join(cmd[1:10])), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:                                   # hand the ranks' stdout through as it comes; remember the JSON line
        sys.stdout.write(ln); sys.stdout.flush()
        if ln.startswith('{'):
            line = ln
    rc = proc.wait()
    if rc != 0:
        raise SystemExit(rc)
    return json.loads(line) if line else None


def main(argv=None):
    args = parse(argv)
    if args.dispatcher:
        return dispatcher_bench(args)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC for RCCL / device-buffer sharing: before the HIP runtime comes up
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')             # one hardware queue per stream of the window lanes (realtime_yukarin_amd/_lib.py)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # started plainly (`python3 bench.py --gpus N`, the form of the driver's 1-GPU command): launch the N ranks ourselves, exactly as
        # the driver's multi-GPU form does, and hand their output through (rank 0 prints the one JSON line)
        return self_launch(args.gpus, list(sys.argv[1:] if argv is None else argv))
    use_dist = world > 1 or args.force_dist
    emu = args.emulator
    model = args.model or ('SYN-8' if emu else 'SYN-64')

    from realtime_yukarin_amd import _lib, engine, sptk, synth
    from realtime_yukarin_amd import dist as rdist
    from realtime_yukarin_amd.netspec import flops as net_flops, pad_frames
    from realtime_yukarin_amd.weights import synthetic_params

    torch = None
    if not emu or (use_dist and args.comm == 'torch'):
        import torch                                            # before the first HIP context: both then share one HIP runtime
    if emu:
        from realtime_yukarin_amd import build
        ctx = engine.Context(0, _lib.Ry355Lib(build.build_emu()))
        dev = None
    else:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs an MI355X: no HIP device visible (there is no CPU path)')
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
        ctx = engine.get_context(local_rank)

    # ---- process group: one rank per GPU; weights travel once, rank 0 -> everyone, over RCCL
    comm = None
    if use_dist:
        if args.comm == 'native':
            comm = rdist.NativeComm(ctx, rank, world)
        else:
            comm = rdist.TorchComm('gloo' if emu else 'nccl', rank, world, dev)
    N, Wn = args.frames, args.windows
    extra = args.extra_frames if args.extra_frames is not None else (100 if N == 300 else 0)
    T = N + pad_frames(N)
    d1, d2 = synth.model_descs(model)
    P1 = synthetic_params(d1, synth.SEED_STAGE1) if rank == 0 else None
    P2 = synthetic_params(d2, synth.SEED_STAGE2) if rank == 0 else None
    if comm is not None:
        net1 = comm.broadcast_net(ctx, d1, P1)
        net2 = comm.broadcast_net(ctx, d2, P2, width=synth.FFT_BINS - 1)
    else:
        from realtime_yukarin_amd.weights import flatten_params
        net1 = engine.Net(ctx, d1, flatten_params(d1, P1))
        net2 = engine.Net(ctx, d2, flatten_params(d2, P2), width=synth.FFT_BINS - 1)
    if args.dtype != 'f32':
        net2.set_dtype(args.dtype)
    mtx = sptk.mc2sp_matrix(d1.out_ch - 1, sptk.mcepalpha(16000), 2 * (synth.FFT_BINS - 1))
    core = engine.VcCore(net1, net2, mtx, lanes=args.lanes)

    def sync_all():
        ctx.sync()
        if torch is not None and not emu:
            torch.cuda.synchronize()

    # ---- synthetic windows, resident in HBM before the timed region (different data per rank).  NW DISTINCT windows take turns (nine:
    # each of the six ring slots of the window call sees three different ones).  Two sets over the same inputs:
    #   'all'   every frame effective (the workload `value` is quoted on);
    #   'mixed' one window in three with a silent stretch, as the silence gate of a live stream leaves it (voice_changer.py:27-37): fewer
    #           effective frames go through stage 1 -- another padded length, i.e. another stage-1 launch plan, and a frame count the
    #           captured stage-1 graphs of a ring slot were not taken for -- and are scattered back into the silent block.
    NW = 9
    xs_host = synth.stage1_input(N, max(NW, Wn), seed=synth.SEED_INPUT + 10 * rank)
    rows_host = numpy.arange(N, dtype=numpy.int32)
    silent_len = {0: max(1, N * 2 // 15), 4: max(1, N // 3), 8: max(1, N * 3 // 5)}        # 300 frames: 260 / 200 / 120 effective -> 384 / 256 / 128 padded
    wsets = {'all': [], 'mixed': []}
    for w in range(max(NW, Wn)):
        for kind in ('all', 'mixed'):
            eff = numpy.ones(N, bool)
            if kind == 'mixed' and w in silent_len:
                a = (N - silent_len[w]) // 2
                eff[a:a + silent_len[w]] = False
            rows = numpy.nonzero(eff)[0].astype(numpy.int32)
            if kind == 'mixed' and not (w in silent_len):
                wsets[kind].append(wsets['all'][w])                                          # the same device buffers
                continue
            dx = ctx.dev_alloc(len(rows) * d1.in_ch); ctx.dev_upload(dx, xs_host[w][eff])
            dr = ctx.dev_alloc(len(rows)); ctx.dev_upload(dr, rows)
            wsets[kind].append(dict(d_x=dx, d_rows=dr, n_eff=int(len(rows)), eff=eff, w=w))
    d_x = [q['d_x'] for q in wsets['all']]
    d_rows = wsets['all'][0]['d_rows']
    # windows that are in flight together write their own result blocks: one block per ring slot (six up to three lanes, else two per lane)
    NB = core.ring * Wn
    d_mc = [ctx.dev_alloc(N * d1.out_ch) for _ in range(NB)]
    d_sp = [ctx.dev_alloc(N * synth.FFT_BINS) for _ in range(NB)]
    turn = {'n': 0, 'w0': 0, 'set': 'all', 'win': 0, 'w0_win': 0}
    sync_all()

    if args.profile_only:
        d_in = ctx.dev_alloc(Wn * T * (synth.FFT_BINS - 1)); d_out = ctx.dev_alloc(Wn * T * (synth.FFT_BINS - 1))
        ctx.dev_upload(d_in, numpy.random.default_rng(0).normal(size=Wn * T * (synth.FFT_BINS - 1)).astype('f4'))
        for _ in range(3):
            net2.forward_device(d_in, d_out, Wn, T)
        ctx.sync()
        st = net2.profile(Wn, T, args.profile_reps)
        if args.layers_out:
            with open(args.layers_out, 'w') as f:
                for q in st:
                    f.write('%-12s %-26s grid=%-10s %9.2f us %8.2f TFLOP/s\n' % (
                        q['layer'], q['name'], 'x'.join(map(str, q['grid'])), q['ms'] * 1e3, q['flops'] / max(q['ms'], 1e-9) / 1e9))
        fam = {}
        for q in st:
            f = fam.setdefault(q['name'], [0.0, 0.0])
            f[0] += q['ms']; f[1] += q['flops']
        print('total %.1fus | ' % (sum(q['ms'] for q in st) * 1e3) + ' | '.join('%s %.1fus %.1fTF' % (k, v[0] * 1e3, v[1] / max(v[0], 1e-9) / 1e9) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])[:5]), flush=True)
        return None

    def step():
        ws = wsets[turn['set']]
        for w in range(Wn):
            k = turn['n'] % NB
            turn['n'] += 1
            q = ws[turn['win'] % len(ws)]
            if w == 0:
                turn['w0'], turn['w0_win'] = k, q['w']
            turn['win'] += 1
            core.enqueue_device(q['d_x'], q['d_rows'], q['n_eff'], N, d_mc[k], d_sp[k], SP_FLOOR)

    def fence():
        sync_all()
        if comm is not None:
            comm.barrier()
        sync_all()

    primed = {'done': False}
    import gc
    import math
    gc_log, gc_t = [], [0.0]

    def gc_note(phase, info):                   # (generation, milliseconds) of every collection, so that a slow bracket can be attributed
        if phase == 'start':
            gc_t[0] = time.perf_counter()
        else:
            gc_log.append((info['generation'], round((time.perf_counter() - gc_t[0]) * 1e3, 2)))
    gc.callbacks.append(gc_note)
    n_prime = 0 if emu else min(144, NB * NW // math.gcd(NB, NW))         # every (result block, window) pair once: 18 with six ring slots and nine windows

    def bracket():
        """ONE timed bracket of exactly K steps: (wall seconds, max over the ranks; seconds to enqueue the steps; device milliseconds between HIP
        events on rank 0).  Opening: barrier + synchronize (fence), then four more untimed steps and a LOCAL synchronize -- the barrier leaves
        the chip idle for as long as the slowest rank and the collective take, and what runs right behind it runs slower for a millisecond or two
        (measured: 40 steps at 1.13 -> 1.19 ms per step after a 10 ms pause, with or without torch.distributed; behind a torch.distributed
        barrier 1.173 / 1.145 / 1.137 ms with 0 / 2 / 8 such steps; queueing the steps UNDER the barrier instead does not help: 1.179).
        The clock starts microseconds after the chip last worked, the ranks as aligned as the barrier left them four steps earlier."""
        fence()
        for _ in range(0 if emu else 4):
            step()
        sync_all()
        gc.disable()                            # (re-enabled below) a full collection is 34-38 ms with torch imported: two brackets' worth of chip time
        ctx.timer_start()                       # every stream is idle: an event on the context stream, nothing joins the lanes
        n_gc = len(gc_log)
        t0 = time.perf_counter()
        if os.environ.get('BENCH_DEBUG_FENCE'):
            t_prev, worst = t0, (0.0, -1)
            for i in range(args.steps):
                step()
                t_now = time.perf_counter()
                worst = max(worst, (t_now - t_prev, i)); t_prev = t_now
            print('rank %d: slowest enqueue of the bracket: step %d, %.3f ms; garbage collections inside: %s'
                  % (rank, worst[1], worst[0] * 1e3, gc_log[n_gc:]), file=sys.stderr, flush=True)
        else:
            for _ in range(args.steps):
                step()
        t_enq = time.perf_counter() - t0
        sync_all()                              # this rank's K steps are done: its clock stops here; the ranks started together (fence above) and
        el = time.perf_counter() - t0           # the job time is the maximum over the ranks (comm.max below) -- the closing barrier itself is not work
        # device-side stamp AFTER the local synchronize: ry_timer_stop while the lanes still have work queued (an event record on every
        # predictor stream plus cross-stream waits) was measured to cost the two lanes their overlap for the whole run (1.33 vs 1.16 ms)
        dms = ctx.timer_stop()
        gc.enable()
        fence()
        if os.environ.get('BENCH_DEBUG_FENCE'):
            print('rank %d: %.3f ms for the steps (%.3f ms to enqueue them, %.3f ms between device events), %.3f ms with the closing barrier'
                  % (rank, el * 1e3, t_enq * 1e3, dms, (time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
        if comm is not None:
            el = comm.max(el)
        return el, t_enq, dms

    def timed(repeats=1):
        """W untimed warm-up steps, then `repeats` brackets of exactly K steps each, back to back.  One bracket of the default K is some tens of
        milliseconds of chip time: a single one is at the mercy of anything else that touches the box in that instant (round 3's driver run
        landed on 2.31 ms per step where every other run of the same binary measured 1.13), so the headline is the MEDIAN bracket and every
        bracket is reported."""
        # launch plans and captured graphs of every ring slot are built before the warm-up (one-off set-up, like loading the weights)
        if not primed['done']:
            for _ in range(n_prime):
                step()
            primed['done'] = True
        # Python's cyclic collector runs on allocation counts: a generation-2 pass (34-38 ms here, torch's object graph) lands in whichever
        # bracket the script's own history puts it -- round 3's single bracket, profiles/r04/driver_cmd.txt.  Everything allocated during
        # set-up is collected once and frozen (moved out of the collector's sight); inside a bracket the collector is off (`bracket`).
        # BEFORE the warm-up: the collection itself idles the chip for those 35 ms, and what runs behind a pause runs slower for a while.
        gc.collect(); gc.freeze()
        for _ in range(args.warmup):
            step()
        return [bracket() for _ in range(max(1, repeats))]

    def median_of(brs):
        els = sorted(b[0] for b in brs)
        return els[(len(els) - 1) // 2]                                   # the lower middle for an even count: never an average of two brackets

    brs = timed(args.repeats)
    elapsed = median_of(brs)
    dev_ms = [b[2] for b in brs if b[0] == elapsed][0]
    sp_gpu = numpy.empty((N, synth.FFT_BINS), numpy.float32); ctx.dev_download(d_sp[turn['w0']], sp_gpu)      # window 0 of the last timed step
    mc_gpu = numpy.empty((N, d1.out_ch), numpy.float32); ctx.dev_download(d_mc[turn['w0']], mc_gpu)
    x_checked = xs_host[turn['w0_win']]                                                                        # ... which converted this input
    assert numpy.isfinite(sp_gpu).all() and numpy.isfinite(mc_gpu).all() and (sp_gpu > 0).all()
    # the same K steps over the 'mixed' set (one window in three cut by the silence gate); every rank runs it (the fences hold barriers)
    turn['set'] = 'mixed'; primed['done'] = False
    elapsed_mixed = median_of(timed(min(3, args.repeats)))
    # a gated window of that region for the parity check: run the shortest one once more, synchronously, into a block of its own
    qm = wsets['mixed'][8 % len(wsets['mixed'])]
    sync_all(); core.enqueue_device(qm['d_x'], qm['d_rows'], qm['n_eff'], N, d_mc[0], d_sp[0], SP_FLOOR); sync_all()
    sp_gated = numpy.empty((N, synth.FFT_BINS), numpy.float32); ctx.dev_download(d_sp[0], sp_gated)
    mc_gated = numpy.empty((N, d1.out_ch), numpy.float32); ctx.dev_download(d_mc[0], mc_gated)
    assert not mc_gated[~qm['eff']].any() and numpy.isfinite(sp_gated).all()
    turn['set'] = 'all'; primed['done'] = False
    for _ in range(n_prime):
        step()
    sync_all(); primed['done'] = True

    frames_total = world * Wn * N * args.steps
    value = frames_total / elapsed
    ms_step = elapsed / args.steps * 1e3
    ms_window = ms_step / Wn
    out = {
        'metric': 'acoustic frames/s (stage1+stage2 fwd) @16kHz/5ms',
        'value': round(value, 1), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_step, 4), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic' if not emu else 'synthetic (EMULATOR: CPU test mode, the numbers mean nothing)',
        # `value` / `ms_per_step` come from the MEDIAN of `repeats` brackets of exactly K steps each (same fences); every bracket is listed:
        # wall ms (max over the ranks), ms the host needed to enqueue the K steps, ms between HIP events on rank 0
        'device_ms_per_step_rank0': round(dev_ms / args.steps, 4),
        'repeats': len(brs), 'value_from': 'median bracket',
        'brackets': [{'wall_ms': round(b[0] * 1e3, 3), 'enq_ms': round(b[1] * 1e3, 3), 'dev_ms': round(b[2], 3)} for b in brs],
        'spread': round((max(b[0] for b in brs) - min(b[0] for b in brs)) / elapsed, 4),
        'slow_brackets': [i for i, b in enumerate(brs) if b[0] > 1.2 * elapsed],
        'gc': 'heap collected + frozen after set-up, collector off inside a bracket (a generation-2 pass is 34-38 ms: profiles/r04/driver_cmd.txt)',
        # every frame handed to convert counts in `value`; ConvertStream keeps only the buffer in the middle of a
        # window with extra_time (convert_stream.py:40-42): effective x real-time = buffer_time / t_wall
        'x_realtime': round(value * 0.005, 1), 'x_realtime_per_gpu': round(value * 0.005 / world, 1),
        'effective_x_realtime': round((N - 2 * extra) * 0.005 / (ms_window * 1e-3), 1),
        'effective_x_realtime_note': '%d of the %d frames of a window are overlap context that ConvertStream discards; buffer_time %.2f s / %.4f ms per window'
                                     % (2 * extra, N, (N - 2 * extra) * 0.005, ms_window),
        'step': 'chained device-resident core (ry_vc_enqueue_device): stage-1 -> combine_silent -> mc2sp + 1e-16 -> stage-2, two windows in flight; '
                '%d distinct windows take turns, every frame effective' % NW,
        'mixed_stream': {'value': round(frames_total / elapsed_mixed, 1), 'unit': 'frames/s', 'ms_per_step': round(elapsed_mixed / args.steps * 1e3, 4),
                         'steps': args.steps, 'effective_frames_of_the_gated_windows': [int(q['n_eff']) for q in wsets['mixed'] if q['n_eff'] < N],
                         'note': 'the same K steps, the same bracket, over the same %d windows with one in three cut by the silence gate (a silent '
                                 'stretch in the middle): stage 1 converts only the effective frames -- another padded length and launch plan, frame '
                                 'counts that change from window to window on every ring slot -- and combine_silent scatters them back; `value` still '
                                 'counts every frame handed to convert' % NW},
        'comm': None if comm is None else comm.kind, 'comm_ranks': None if comm is None else comm.ranks_seen,
        'config': {'workload': 'BASELINE config #3: stage-1 + stage-2 SR forward, buffer_time 0.5 s + 2x0.5 s convert_extra_time '
                               '@16 kHz / 5 ms -> %d real frames (%d padded) per window, %d window(s) per GPU per step, %s random-init weights'
                               % (N, T, Wn, model),
                   'model': model, 'frames': N, 'padded_frames': T, 'windows_per_gpu': Wn,
                   'lanes_per_gpu': core.lanes,
                   'parallelism': 'chunk-dp%d (independent windows, RCCL weight broadcast at init, no steady-state collective); %d window lane(s) per GPU' % (world, core.lanes)},
    }

    def convert_now(widx):
        """Window `widx` of the all-effective set through the current settings, synchronously, into result block 0: (mc, sp) on the host."""
        q = wsets['all'][widx]
        sync_all(); core.enqueue_device(q['d_x'], q['d_rows'], q['n_eff'], N, d_mc[0], d_sp[0], SP_FLOOR); sync_all()
        sp_ = numpy.empty((N, synth.FFT_BINS), numpy.float32); ctx.dev_download(d_sp[0], sp_)
        mc_ = numpy.empty((N, d1.out_ch), numpy.float32); ctx.dev_download(d_mc[0], mc_)
        return mc_, sp_

    def time_only(fn, reps=20):
        for _ in range(3):
            fn()
        ctx.sync(); ctx.timer_start()
        for _ in range(reps):
            fn()
        return ctx.timer_stop() / reps

    if rank == 0 and not emu:
        d_y1 = ctx.dev_alloc(N * d1.out_ch); d_s2in = ctx.dev_alloc(N * synth.FFT_BINS); d_s2out = ctx.dev_alloc(N * synth.FFT_BINS)
        ctx.dev_upload(d_s2in, synth.stage2_input(N)[0])
        s1_ms = time_only(lambda: net1.convert_device(d_x[0], d_y1, 1, N))
        s2_ms = time_only(lambda: net2.convert_device(d_s2in, d_s2out, 1, N))
        per = []                                                          # one window at a time, a sync after each: no overlap (median of 20)
        for i in range(23):
            tq = time.perf_counter(); step(); ctx.sync(); per.append((time.perf_counter() - tq) * 1e3 / Wn)
        chain_ms = sorted(per[3:])[10]
        out['graph_replay_ms'] = {'stage1_alone': round(s1_ms, 4), 'stage2_alone': round(s2_ms, 4), 'chain_one_window_synced': round(chain_ms, 4)}
        # live per-launch profile (HIP events around every launch of both predictors) for the roofline objects
        st2 = net2.profile(1, N, args.profile_reps, window=True)        # launch by launch, the convert wrapper on one window: what the step runs
        st1 = net1.profile(1, N, args.profile_reps, window=True)

        if not args.no_extras:
            # stage-1 with its filters evicted (cold next to warm): 512 MiB of scratch is rewritten before every replay
            scratch = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
            cold = []
            for i in range(6):
                scratch.fill_(i & 1); sync_all()
                ctx.timer_start()
                net1.convert_device(d_x[0], d_y1, 1, N)
                cold.append(ctx.timer_stop())
            out['graph_replay_ms']['stage1_alone_cold'] = round(sorted(cold[1:])[len(cold[1:]) // 2], 4)
            del scratch
            # host windows (PCIe inclusive; never the headline): one at a time, and a stream with windows in flight through the pinned ring
            xh = xs_host[0]; eff = numpy.ones(N, bool)
            for _ in range(12):                                   # every ring slot has its launch plans and graphs
                core.convert(xh, eff)
            th = time.perf_counter()
            for _ in range(20):
                core.convert(xh, eff)
            host_ms = (time.perf_counter() - th) / 20 * 1e3
            for _ in core.convert_stream([(xh, eff)] * 12, depth=6):
                pass
            th = time.perf_counter()
            for _ in core.convert_stream([(xh, eff)] * 60, depth=6):
                pass
            stream_ms = (time.perf_counter() - th) / 60 * 1e3
            # the same call with the silence gate on the device (raw wave + all frames up; ry_vc_submit_wave)
            from realtime_yukarin_amd import gate
            wv32 = (0.1 * numpy.random.default_rng(1).normal(size=N * 80)).astype(numpy.float32)
            p_eff, p_all = gate.thresholds(60)
            for _ in range(12):
                core.wait_wave(core.submit_wave(wv32, 80, 1024, p_eff, p_all, xh))
            th = time.perf_counter()
            for _ in range(20):
                core.wait_wave(core.submit_wave(wv32, 80, 1024, p_eff, p_all, xh))
            gated_ms = (time.perf_counter() - th) / 20 * 1e3
            # what a LIVE stream sees (one buffer every buffer_time, no backlog): one synchronous call per window.  The mirror worker announces the
            # overlap frames ConvertStream.process picks away (worker.convert_worker: discard = (pad, pad)); and the gate makes the number of
            # effective frames change from window to window on the same ring buffers (stage-1 graphs are taken per frame count)
            live = {}
            if extra > 0:
                core.set_discard(extra, extra)
                for _ in range(12):
                    core.convert(xh, eff)
                th = time.perf_counter()
                for _ in range(20):
                    core.convert(xh, eff)
                live['call_ms_per_window_with_discard_hint'] = round((time.perf_counter() - th) / 20 * 1e3, 4)
                core.set_discard(0, 0)
            effs = []
            for i in range(6):
                e = numpy.ones(N, bool); e[20 + 7 * i:20 + 7 * i + 11 + 9 * i] = False
                effs.append(e)
            for i in range(24):
                core.convert(xh[effs[i % 6]], effs[i % 6])
            th = time.perf_counter()
            for i in range(24):
                core.convert(xh[effs[i % 6]], effs[i % 6])
            live['call_ms_per_window_varying_effective_frames'] = round((time.perf_counter() - th) / 24 * 1e3, 4)
            for _ in range(12):
                core.convert(xh, eff)
            out['host_path'] = {'call_ms_per_window': round(host_ms, 4), 'stream_ms_per_window': round(stream_ms, 4), **live,
                                'call_with_device_gate_ms_per_window': round(gated_ms, 4),
                                'stream_frames_per_s': round(N / (stream_ms * 1e-3), 1),
                                'note': 'host arrays in, host arrays out through the pinned ring of ry_vc_submit / ry_vc_wait (PCIe inclusive): one window '
                                        'at a time, and a stream with six windows in flight'}
            # 8 windows per stage-1 call: the regime in which the filters are amortised (bytes per frame / 8)
            d_x8 = ctx.dev_alloc(8 * N * d1.in_ch); d_y8 = ctx.dev_alloc(8 * N * d1.out_ch)
            ctx.dev_upload(d_x8, synth.stage1_input(N, 8))
            s1x8_ms = time_only(lambda: net1.convert_device(d_x8, d_y8, 8, N))
            out['stage1_batch8'] = {'ms_per_call': round(s1x8_ms, 4), 'frames_per_s': round(8 * N / (s1x8_ms * 1e-3), 1)}
            # several windows per stage-2 call (independent streams served by one GPU, or run.py's backlog): the filters of the bottom
            # layers are streamed once per call and every grid is many rounds of workgroups, so the dead-row crop pays on every decoder layer
            out['stage2_batch'] = {}
            for bw in (4, 8):
                d_b_in = ctx.dev_alloc(bw * N * synth.FFT_BINS); d_b_out = ctx.dev_alloc(bw * N * synth.FFT_BINS)
                s2x = synth.stage2_input(N)[0]
                ctx.dev_upload(d_b_in, numpy.ascontiguousarray(numpy.broadcast_to(s2x, (bw,) + s2x.shape)))
                b_ms = time_only(lambda: net2.convert_device(d_b_in, d_b_out, bw, N), reps=10)
                yb = numpy.empty((bw,) + s2x.shape, numpy.float32); ctx.dev_download(d_b_out, yb)
                ref1 = numpy.empty(s2x.shape, numpy.float32); ctx.dev_download(d_s2out, ref1)
                out['stage2_batch'][str(bw)] = {'ms_per_call': round(b_ms, 4), 'ms_per_window': round(b_ms / bw, 4), 'frames_per_s': round(bw * N / (b_ms * 1e-3), 1),
                                                'mfma_frac_padded_flops': round(bw * net_flops(d2, T, synth.FFT_BINS - 1) / (b_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TF, 4),
                                                'max_rel_diff_vs_single_window': float(numpy.abs(yb.astype(numpy.float64) / ref1 - 1.0).max())}
                ctx.dev_free(d_b_in); ctx.dev_free(d_b_out)
            # the whole chained window call for 8 windows at once (ry_vc_enqueue_device_batch): stage 1 as one batch, the hop on the device, stage 2 as one batch
            for bw in [int(v) for v in args.chained_batch.split(',') if v]:
                xb = synth.stage1_input(N, max(NW, Wn, bw), seed=synth.SEED_INPUT + 10 * rank)[:bw].copy()
                xb[0] = x_checked                                                        # window 0 is the window the timed step converted last
                d_bx = ctx.dev_alloc(bw * N * d1.in_ch); d_br = ctx.dev_alloc(bw * N)
                d_bmc = ctx.dev_alloc(bw * N * d1.out_ch); d_bsp = ctx.dev_alloc(bw * N * synth.FFT_BINS)
                ctx.dev_upload(d_bx, xb); ctx.dev_upload(d_br, numpy.tile(rows_host, bw))
                cb_ms = time_only(lambda: core.enqueue_device_batch(d_bx, d_br, [N] * bw, N, d_bmc, d_bsp, SP_FLOOR), reps=10)
                spb = numpy.empty((bw, N, synth.FFT_BINS), numpy.float32); ctx.dev_download(d_bsp, spb)
                out['chained_batch%d' % bw] = {'windows_per_call': bw, 'ms_per_call': round(cb_ms, 4), 'ms_per_window': round(cb_ms / bw, 4),
                                         'frames_per_s': round(bw * N / (cb_ms * 1e-3), 1), 'x_realtime': round(bw * N / (cb_ms * 1e-3) * 0.005, 1),
                                         'effective_x_realtime': round(bw * (N - 2 * extra) * 0.005 / (cb_ms * 1e-3), 1),
                                         'max_rel_diff_window0_vs_timed_step': float(numpy.abs(spb[0].astype(numpy.float64) / sp_gpu - 1.0).max()),
                                         'note': 'the same chained core on %d independent windows per call (streams served side by side / a backlog); not the headline: '
                                                 'the headline step is one 0.5 s buffer at a time, as the reference converts' % bw}
                for q in (d_bx, d_br, d_bmc, d_bsp):
                    ctx.dev_free(q)
            # the same step when the caller announces the frames it will throw away (ConvertStream.process keeps the buffer in the middle of
            # the window it converted; worker.convert_worker passes its pad): stage 2 computes the kept rows only.  NOT the headline -- the
            # headline returns every frame of every window -- but what a live stream built on the mirror worker runs.
            if extra > 0:
                core.set_discard(extra, extra)
                for _ in range(12):
                    step()
                sync_all()
                td = time.perf_counter()
                for _ in range(args.steps):
                    step()
                sync_all()
                d_ms = (time.perf_counter() - td) / args.steps / Wn * 1e3
                _, spd = convert_now(turn['w0_win'])                               # a window under the hint ...
                for _ in core.convert_stream([(xh, eff)] * 12, depth=6):          # the same through the pinned ring, host arrays in and out
                    pass
                th = time.perf_counter()
                for _ in core.convert_stream([(xh, eff)] * 60, depth=6):
                    pass
                d_stream_ms = (time.perf_counter() - th) / 60 * 1e3
                core.set_discard(0, 0)
                _, spf = convert_now(turn['w0_win'])                               # ... and the same window in full
                for _ in range(12):
                    step()
                sync_all()
                out['discard_hint'] = {'discard_front_back': [extra, extra], 'ms_per_window': round(d_ms, 4),
                                       'effective_x_realtime': round((N - 2 * extra) * 0.005 / (d_ms * 1e-3), 1),
                                       'host_stream_ms_per_window': round(d_stream_ms, 4),
                                       'kept_rows_bit_identical_to_the_full_step': bool(numpy.array_equal(spd[extra:N - extra], spf[extra:N - extra]) and spf[extra:N - extra].min() > 0),
                                       'note': 'ry_vc_set_discard: the decoder layers of stage 2 run on the row range the kept frames depend on; '
                                               'encoder and bottom of the U-Net whole; stage 1 whole; not the headline (the headline returns all frames)'}
            # the silence gate on this box's host (is it worth a kernel?)
            compat_dir = ROOT / 'realtime_yukarin_amd' / 'compat'
            sys.path.insert(0, str(compat_dir))
            from yukarin.wave import Wave
            wv = Wave((0.1 * numpy.random.default_rng(1).normal(size=N * 80)).astype(numpy.float32), 16000)
            wv.get_effective_frame(60, 1024, 5)
            tg = time.perf_counter()
            for _ in range(20):
                wv.get_effective_frame(60, 1024, 5)
            gate_ms = (time.perf_counter() - tg) / 20 * 1e3
            out['silence_gate_host'] = {'ms_per_window': round(gate_ms, 4), 'fraction_of_window': round(gate_ms / ms_window, 4),
                                        'note': 'separate_effective mask (numpy, one host thread); it overlaps the GPU work of the previous window'}

        # BASELINE config #3 / #4 literally: buffer_time 0.5 s and NO extra context = 100-frame windows (/root/reference/config.yaml:6,14 with
        # convert_extra_time 0) -- what a low-latency live stream converts.  Same core, same bracket procedure; every grid of stage 2 is below
        # one round of workgroups at 128 padded frames and the weight-streaming bottom of the U-Net costs what it costs at any window size.
        if not args.no_extras and N > 100 and world == 1:
            Ns = 100
            xs_s = synth.stage1_input(Ns, NW, seed=synth.SEED_INPUT + 99)
            d_xs = []
            for w in range(NW):
                q = ctx.dev_alloc(Ns * d1.in_ch); ctx.dev_upload(q, xs_s[w]); d_xs.append(q)
            d_rows_s = ctx.dev_alloc(Ns); ctx.dev_upload(d_rows_s, numpy.arange(Ns, dtype=numpy.int32))
            ks = {'n': 0}

            def step_s():
                k = ks['n'] % NB; ks['n'] += 1
                core.enqueue_device(d_xs[ks['n'] % NW], d_rows_s, Ns, Ns, d_mc[k], d_sp[k], SP_FLOOR)      # (the 300-frame result blocks are large enough)
            for _ in range(n_prime + 6):
                step_s()
            sync_all()
            els = []
            for _ in range(3):
                for _ in range(4):
                    step_s()
                sync_all(); gc.disable(); ts = time.perf_counter()
                for _ in range(2 * args.steps):
                    step_s()
                sync_all(); els.append((time.perf_counter() - ts) / (2 * args.steps)); gc.enable()
            ms_s = sorted(els)[1] * 1e3
            d_si = ctx.dev_alloc(Ns * synth.FFT_BINS); d_so = ctx.dev_alloc(Ns * synth.FFT_BINS)
            ctx.dev_upload(d_si, synth.stage2_input(Ns)[0])
            s2s_ms = time_only(lambda: net2.convert_device(d_si, d_so, 1, Ns))
            run_s = sum(q['flops'] for q in net2.profile(1, Ns, 1, window=True))
            per_s = []
            for i in range(13):
                tq = time.perf_counter(); step_s(); ctx.sync(); per_s.append((time.perf_counter() - tq) * 1e3)
            out['small_window'] = {'frames': Ns, 'padded_frames': Ns + pad_frames(Ns), 'value': round(Ns / (ms_s * 1e-3), 1), 'unit': 'frames/s',
                                   'ms_per_window': round(ms_s, 4), 'x_realtime': round(Ns / (ms_s * 1e-3) * 0.005, 1),
                                   'chain_one_window_synced_ms': round(sorted(per_s[3:])[5], 4), 'stage2_alone_ms': round(s2s_ms, 4),
                                   'stage2_forward_frac': round(run_s / (s2s_ms * 1e-3) / 1e12 / (F32_MFMA_PEAK_TF if args.dtype == 'f32' else 2500.0), 4),
                                   'note': 'BASELINE config #3 / #4 core: 100-frame windows (buffer_time 0.5 s, no extra context), two lanes; median of three brackets'}
            for q in d_xs + [d_rows_s, d_si, d_so]:
                ctx.dev_free(q)
            for _ in range(n_prime):                                                 # back to the headline windows on every ring slot
                step()
            sync_all()

        # the same K steps with stage-2 in split-bf16 mode -- reported BESIDE the exact-fp32 headline, never as `value` of an f32 run
        if args.dtype == 'f32' and not args.no_split_bf16 and not args.no_extras and world == 1:
            net2.set_dtype('bf16x3')
            primed['done'] = False
            el3 = median_of(timed(min(3, args.repeats)))
            _, sp3 = convert_now(turn['w0_win'])                                   # one window in split-bf16 mode ...
            net2.set_dtype('f32')
            _, sp3_f32 = convert_now(turn['w0_win'])                               # ... and the same window on the exact fp32 path
            step(); sync_all()
            out['split_bf16'] = {'dtype': 'bf16x3', 'value': round(world * Wn * N * args.steps / el3, 1), 'unit': 'frames/s',
                                 'ms_per_step': round(el3 / args.steps * 1e3, 4),
                                 'max_rel_diff_vs_f32_path': float(numpy.abs(sp3.astype(numpy.float64) / sp3_f32 - 1.0).max()), 'parity_bar': 1e-4,
                                 'note': 'stage-2 MFMA-bound layers as hi*hi + lo*hi + hi*lo bf16 products, fp32 accumulate; opt-in (--dtype bf16x3); the headline is exact fp32'}

        # ---- roofline of the dominant kernel
        if args.layers_out:
            with open(args.layers_out, 'w') as f:
                for tag, st in (('stage1', st1), ('stage2', st2)):
                    for s in st:
                        f.write('%-7s %-12s %-40s grid=%-16s %9.2f us %8.2f TFLOP/s %9.1f GB/s\n' % (
                            tag, s['layer'], s['name'], 'x'.join(map(str, s['grid'])), s['ms'] * 1e3,
                            s['flops'] / max(s['ms'], 1e-9) / 1e9, s['bytes'] / max(s['ms'], 1e-9) / 1e6))
        fam = {}
        for s in st2 + st1:
            f = fam.setdefault(s['name'], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, exec=0.0))
            f['ms'] += s['ms']; f['flops'] += s['flops']; f['bytes'] += s['bytes']; f['launches'] += 1; f['exec'] += s.get('flops_exec', s['flops'])
        # dominant kernel = the family with the most time in the stage that bounds the step (stage 2; stage 1 overlaps on its own stream)
        st2_names = set(s['name'] for s in st2)
        dname, dv = max(((k, v) for k, v in fam.items() if k in st2_names), key=lambda kv: kv[1]['ms'])
        # `achieved` / `frac` count the matrix-pipe FLOPs the launches EXECUTE (round 6: the Winograd F(2x2, 2x2) kernels run 9 / 16 of the direct
        # convolution's products -- ry_kernel_stat.flops_exec); the SURVEY 8(d) direct-convolution figure stays beside it as `frac_algorithmic`
        ach = dv['exec'] / (dv['ms'] * 1e-3) / 1e12
        pmc, pmc_file = pmc_table()
        rpf, rpf_file = rocprof_table()
        targs = dname[dname.find('<') + 1:-1].split(',') if dname.startswith('ry_igemm_ldsdma<') else []
        is_bf16 = len(targs) > 5 and targs[5] == 'true'
        peak_tf = 2500.0 if is_bf16 else F32_MFMA_PEAK_TF
        # `frac` comes from the rocprofv3 average of the SAME source (profiles/<..>kernel_stats.txt, matched by the hash of csrc/), so that it
        # can be recomputed from profiles/; the live HIP-event figure of this very run stays beside it as `frac_events`
        ach_ev = ach
        rp_us = rpf.get(dname.replace(' ', ''))
        if rp_us:
            ach = dv['exec'] / dv['launches'] / (rp_us * 1e-6) / 1e12
        out['roofline'] = {'kernel': dname, 'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak_tf, 'unit': 'TFLOP/s',
                           'frac': round(ach / peak_tf, 4),
                           'frac_source': ('%s: avg %.3f us per launch (rocprofv3 --kernel-trace --stats, one window at a time); this run: box %s, %s, frac_events' % (rpf_file, rp_us, os.uname().nodename, time.strftime('%Y-%m-%d', time.gmtime()))) if rp_us
                                          else 'HIP events of this run (no rocprofv3 summary of source %s under profiles/)' % source_hash(),
                           'achieved_events': round(ach_ev, 2), 'frac_events': round(ach_ev / peak_tf, 4),      # HIP events around every launch, THIS run
                           'frac_rocprof': round(ach / peak_tf, 4) if rp_us else None,                         # committed rocprofv3 summary of the same source
                           'traffic': pmc.get(dname.replace(' ', ''), (None,))[0],
                           'traffic_unit': 'HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc)',
                           'traffic_source': pmc_file, 'source_hash': source_hash(),
                           'launches': dv['launches'], 'avg_launch_ms': round(dv['ms'] / dv['launches'], 4),
                           'alg_flops_per_launch': dv['flops'] / dv['launches'], 'alg_bytes_per_launch': dv['bytes'] / dv['launches'],
                           'exec_flops_per_launch': dv['exec'] / dv['launches'],
                           'frac_algorithmic': round(ach / peak_tf * dv['flops'] / max(dv['exec'], 1.0), 4),
                           'flops_basis': 'achieved / frac: matrix-pipe FLOPs the launches execute (Winograd F(2x2, 2x2): 9 products per 2 x 2 outputs of a 2 x 2-tap stencil instead of 16); '
                                          'frac_algorithmic: the direct-convolution FLOPs of SURVEY.md 8(d) over the same time'}
        # the other MFMA-bound families of the forward, same arithmetic (rocprofv3 average of this source where there is one, else this run's HIP events):
        # which family carries the most time can change with the plans (round 5: decoder c3 moved from the <..,2,false,1> to the <..,1,false,1> family)
        others = {}
        for k, v in sorted(((k, v) for k, v in fam.items() if k in st2_names and k != dname and k.startswith(('ry_igemm_ldsdma<', 'ry_wino_ldsdma<')) and v['flops'] > 0),
                           key=lambda kv: -kv[1]['ms'])[:4]:
            us = rpf.get(k.replace(' ', '')) or v['ms'] / v['launches'] * 1e3
            others[k] = {'launches': v['launches'], 'avg_launch_us': round(us, 2), 'frac': round(v['exec'] / v['launches'] / (us * 1e-6) / 1e12 / peak_tf, 4),
                         'frac_algorithmic': round(v['flops'] / v['launches'] / (us * 1e-6) / 1e12 / peak_tf, 4)}
        out['roofline']['other_families'] = others
        if args.dtype == 'bf16x3' and is_bf16:
            out['roofline']['mfma_flops_per_alg_flop'] = 3
        alg2 = net_flops(d2, T, synth.FFT_BINS - 1)
        pk2 = F32_MFMA_PEAK_TF if args.dtype == 'f32' else 2500.0
        run2 = sum(s.get('flops_exec', s['flops']) for s in st2)         # what the launches execute: decoder rows that only feed the cropped padding are skipped, Winograd layers run 9 / 16 of their products
        run2_alg = sum(s['flops'] for s in st2)                          # the same rows as direct convolutions (SURVEY.md 8(d) formula)
        out['roofline_stage2_forward'] = {'bound': 'mfma', 'achieved': round(run2 / (s2_ms * 1e-3) / 1e12, 2), 'peak': pk2,
                                          'unit': 'TFLOP/s', 'frac': round(run2 / (s2_ms * 1e-3) / 1e12 / pk2, 4),
                                          'frac_algorithmic': round(run2_alg / (s2_ms * 1e-3) / 1e12 / pk2, 4),
                                          'executed_gflop': round(run2 / 1e9, 3), 'direct_form_gflop': round(run2_alg / 1e9, 3), 'padded_forward_gflop': round(alg2 / 1e9, 3),
                                          'frac_if_all_padded_rows_counted': round(alg2 / (s2_ms * 1e-3) / 1e12 / pk2, 4),
                                          'note': 'FLOPs the stage-2 launches execute / graph replay time of the whole forward (end layers, reduces and launches included); '
                                                  'padded_forward_gflop is the forward over all T padded rows, which the reference computes and then crops'}
        c1 = [s for s in st1 if s['name'].startswith(('ry_c1d_os', 'ry_conv1d_ws'))]
        ms1 = sum(s['ms'] for s in c1); by1 = sum(s['bytes'] for s in c1)
        t1 = [pmc[s['name']][0] for s in c1 if s['name'] in pmc]
        out['roofline_stage1'] = {'kernel': '%s<*> (%d launches)' % (c1[0]['name'].split('<')[0], len(c1)), 'bound': 'hbm',
                                  'achieved': round(by1 / (s1_ms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                  'frac': round(by1 / (s1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  'traffic': sum(t1) if len(t1) == len(c1) else None, 'traffic_unit': 'HBM bytes per forward (sum over the launches)',
                                  'alg_bytes_per_forward': by1, 'kernel_ms_per_forward': round(s1_ms, 4),
                                  'kernel_ms_sum_of_eager_launches': round(ms1, 4),
                                  'frac_graph_warm': round(by1 / (s1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  'note': 'achieved = algorithmic bytes (filters + input + output once) / graph replay time of the whole stage-1 forward'}
        if 'stage1_alone_cold' in out['graph_replay_ms']:
            out['roofline_stage1']['frac_graph_cold'] = round(by1 / (out['graph_replay_ms']['stage1_alone_cold'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if 'stage1_batch8' in out:
            b8 = by1 + 7 * (N * (d1.in_ch + d1.out_ch) * 4)
            out['stage1_batch8']['alg_bytes_per_frame'] = round(b8 / (8 * N), 1)
            out['stage1_batch8']['hbm_frac'] = round(b8 / (out['stage1_batch8']['ms_per_call'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        out['kernels'] = {k: {'ms': round(v['ms'], 4), 'launches': v['launches'], 'tflops': round(v['flops'] / max(v['ms'], 1e-9) / 1e9, 2)}
                          for k, v in sorted(fam.items(), key=lambda kv: -kv[1]['ms'])}
        out['alg'] = {'stage1_gflop': net_flops(d1, T) / 1e9, 'stage2_gflop': alg2 / 1e9}

        # ---- CPU baseline beside it: the oracle's torch/oneDNN restatement of the SAME chained window on this box's host cores
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(args, torch, d1, d2, x_checked, mc_gpu, sp_gpu, N, value,
                                               gated=(xs_host[qm['w']], qm['eff'], mc_gated, sp_gated))
            out['mixed_stream']['gpu_result_of_a_gated_window_vs_cpu_restatement'] = out['cpu_baseline'].pop('gated_window')
    if rank == 0:
        details = args.details_out or str(ROOT / 'gpurun_out' / ('bench_details_%dgpu.json' % world))
        try:
            Path(details).parent.mkdir(parents=True, exist_ok=True)
            Path(details).write_text(json.dumps(out, indent=1))
        except OSError as e:
            details = 'not written (%s)' % e
        print(json.dumps(compact_line(out, details)), flush=True)
    core.close(); net1.close(); net2.close()
    if comm is not None:
        comm.close()
    return out


def compact_line(out, details):
    """The ONE printed line: headline, brackets, roofline objects, CPU baseline and a few secondary figures -- short enough (< 6 KB) that a
    tail of the output still holds `value`, `brackets` and `device_ms_per_step_rank0`.  Everything else (`kernels`, batch / host-path /
    discard measurements, the notes) is in the details file."""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
            'device_ms_per_step_rank0', 'repeats', 'value_from', 'brackets', 'spread', 'slow_brackets', 'gc', 'x_realtime', 'x_realtime_per_gpu',
            'effective_x_realtime', 'comm', 'comm_ranks', 'config', 'graph_replay_ms')
    line = {k: out[k] for k in keep if k in out}

    def pick(src, keys):
        return {k: src[k] for k in keys if k in src}
    if 'mixed_stream' in out:
        line['mixed_stream'] = pick(out['mixed_stream'], ('value', 'unit', 'ms_per_step', 'gpu_result_of_a_gated_window_vs_cpu_restatement'))
    if 'roofline' in out:
        line['roofline'] = pick(out['roofline'], ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'frac_source', 'frac_events', 'frac_rocprof',
                                                  'traffic', 'traffic_source', 'source_hash', 'launches', 'avg_launch_ms', 'alg_flops_per_launch',
                                                  'alg_bytes_per_launch', 'exec_flops_per_launch', 'frac_algorithmic', 'mfma_flops_per_alg_flop', 'other_families'))
    if 'roofline_stage2_forward' in out:
        line['roofline_stage2_forward'] = pick(out['roofline_stage2_forward'], ('bound', 'achieved', 'peak', 'unit', 'frac', 'frac_algorithmic', 'executed_gflop', 'direct_form_gflop', 'padded_forward_gflop'))
    if 'roofline_stage1' in out:
        line['roofline_stage1'] = pick(out['roofline_stage1'], ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'frac_graph_cold', 'traffic',
                                                                'alg_bytes_per_forward', 'kernel_ms_per_forward'))
    if 'cpu_baseline' in out:
        line['cpu_baseline'] = pick(out['cpu_baseline'], ('value', 'unit', 'cores', 'kind', 'sample', 'frames_per_s_by_threads', 'gpu_over_cpu',
                                                          'gpu_result_vs_this_baseline'))
    if 'host_path' in out:
        line['host_path'] = pick(out['host_path'], ('call_ms_per_window', 'call_ms_per_window_with_discard_hint', 'stream_ms_per_window', 'stream_frames_per_s'))
    for k in ('small_window', 'dispatcher', 'split_bf16', 'discard_hint', 'chained_batch8', 'stage1_batch8'):
        if k in out:
            line[k] = {kk: vv for kk, vv in out[k].items() if kk != 'note' and not isinstance(vv, (dict, list))}
    line['details'] = details
    return line


def cpu_baseline(args, torch, d1, d2, x, mc_gpu, sp_gpu, N, gpu_value, gated=None):
    """Bounded sample of the same workload on the host cores: the torch/oneDNN restatement (oracle/torch_ref.py) of stage-1 -> mc2sp ->
    stage-2 on the window the GPU just converted -- all threads, then one thread -- and the GPU result checked against it."""
    from oracle import mc2sp as omc, torch_ref
    from realtime_yukarin_amd import synth
    from realtime_yukarin_amd.weights import synthetic_params
    P1 = synthetic_params(d1, synth.SEED_STAGE1); P2 = synthetic_params(d2, synth.SEED_STAGE2)
    t1n, t2n = torch_ref.TorchUNet(P1), torch_ref.TorchUNet(P2)
    alpha = omc.mcepalpha(16000)

    def chain(xw):
        mc = torch_ref.stage1_convert_core(t1n, xw)
        sp_mid = (omc.mc2sp(mc, alpha, 1024) + SP_FLOOR).astype(numpy.float32)
        return mc, torch_ref.stage2_convert(t2n, sp_mid)
    mc_ref, sp_ref = chain(x)                                  # warm-up + the check
    err_sp = float(numpy.abs(sp_gpu.astype(numpy.float64) / sp_ref - 1).max())
    err_mc = float(numpy.abs(mc_gpu - mc_ref).max() / numpy.abs(mc_ref).max())
    gated_res = None
    if gated is not None:                                      # a window of the 'mixed' region: stage 1 on the effective frames, zeros elsewhere (voice_changer.py:33-37)
        xg, eff, mc_g, sp_g = gated
        mcw = numpy.zeros((N, mc_ref.shape[1]), numpy.float32)
        mcw[eff] = torch_ref.stage1_convert_core(t1n, xg[eff])
        spw = torch_ref.stage2_convert(t2n, (omc.mc2sp(mcw, alpha, 1024) + SP_FLOOR).astype(numpy.float32))
        gated_res = {'effective_frames': int(eff.sum()), 'sp_max_rel': float(numpy.abs(sp_g.astype(numpy.float64) / spw - 1).max()),
                     'mc_max_norm': float(numpy.abs(mc_g - mcw).max() / numpy.abs(mcw).max()), 'bar': 1e-4}
        assert gated_res['sp_max_rel'] < 1e-4 and gated_res['mc_max_norm'] < 1e-4, 'gated window differs from the CPU restatement: %r' % gated_res
    reps, tb = 0, time.perf_counter()
    while True:
        chain(x); reps += 1
        if time.perf_counter() - tb > args.cpu_seconds or reps >= 50:
            break
    cpu_s = (time.perf_counter() - tb) / reps
    nthr = torch.get_num_threads()
    torch.set_num_threads(1)
    t1 = time.perf_counter(); chain(x); one_s = time.perf_counter() - t1
    mid = max(2, min(16, nthr // 2))
    torch.set_num_threads(mid)                                # oneDNN at batch 1 does not scale to every core of a large host: a middle setting beside 1 / all
    chain(x); t1 = time.perf_counter(); chain(x); chain(x); mid_s = (time.perf_counter() - t1) / 2
    torch.set_num_threads(nthr)
    by_threads = {nthr: N / cpu_s, mid: N / mid_s, 1: N / one_s}
    best = max(by_threads, key=lambda k: by_threads[k])     # oneDNN at batch 1 does not scale to every core of a large host: quote the best setting
    res = {'value': round(by_threads[best], 1), 'unit': 'frames/s', 'cores': best, 'kind': 'port',
           'sample': '%d x (stage-1 -> mc2sp -> stage-2 of one %d-frame window) with all %d threads, 2 x with %d, 1 x with one; `value` = the best of the three; '
                     'CPU restatement (torch/oneDNN fp32), not Chainer; host has %d logical cpus' % (reps, N, nthr, mid, os.cpu_count()),
           'frames_per_s_by_threads': {str(k): round(v, 1) for k, v in sorted(by_threads.items())},
           'gpu_over_cpu': round(gpu_value / by_threads[best], 1),
           'gpu_result_vs_this_baseline': {'sp_max_rel': err_sp, 'mc_max_norm': err_mc, 'bar': 1e-4}, 'gated_window': gated_res}
    assert err_sp < 1e-4 and err_mc < 1e-4, 'GPU result of the timed region differs from the CPU restatement: %g %g' % (err_sp, err_mc)
    # BASELINE config #1, the plumbing baseline: check.py's schedule (/root/reference/check.py:96-125) = 5 windows of 1 s + 2 x 1 s
    # extra = 600 frames each, converted one after the other; here with the same restatement, one pass
    x6 = synth.stage1_input(600, 5, seed=synth.SEED_INPUT + 7)
    t6 = time.perf_counter()
    for w in range(5):
        chain(x6[w])
    c1 = time.perf_counter() - t6
    res['config1_check_py_schedule'] = {'cpu_s_for_5_windows_of_600_frames': round(c1, 3), 'input_seconds': 5.0, 'cpu_x_realtime': round(5.0 / c1, 2),
                                        'note': 'convert stage of check.py (5 x 1 s buffers, extra_time 1 s -> 600-frame windows) on the host cores, one pass; '
                                                'the GPU runs the same schedule in `python bench.py --frames 600 --extra-frames 200`'}
    return res


if __name__ == '__main__':
    main()
