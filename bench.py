#!/usr/bin/env python3
"""bench.py -- stage-1 + stage-2 forward throughput of the convert hot path on N MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it under
torch.distributed.run (one rank per GPU, RCCL).  One "step" = one pass of the hot path over one batch of
synthetic windows already resident in HBM: `AcousticConverter.convert` array part (stage-1) followed by
`SuperResolution.convert` (stage-2) for `--windows` windows of `--frames` real frames per GPU
(default: 1 window of 300 frames = buffer_time 0.5 s + 2 x convert_extra_time 0.5 s at 5 ms frames, the
window /root/reference/config.yaml:14 gives BASELINE config #3).  Windows are independent, so N GPUs take
N times the windows (weak scaling) with one RCCL broadcast of the weight blobs at start-up and no
collective in the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--frames', type=int, default=300, help='real frames per window (N)')
    ap.add_argument('--windows', type=int, default=1, help='windows per GPU per step')
    ap.add_argument('--model', default='SYN-64')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16', 'bf16x3'],
                    help="stage-2 MFMA operand type ('bf16' = BASELINE config #5: bf16 filters and activations between the implicit-GEMM layers, fp32 "
                         "accumulation and end layers; 'bf16x3' = split-bf16: every fp32 product as hi*hi + lo*hi + hi*lo on the bf16 pipe, fp32 "
                         "accumulate, results within ~2e-6 of the fp32 path -- DESIGN.md 4.7)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-split-bf16', action='store_true', help="skip the extra 'split_bf16' measurement of an f32 run")
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--profile-reps', type=int, default=5)
    ap.add_argument('--profile-only', action='store_true', help='only print the per-kernel-family profile of stage-2 (tuning aid)')
    ap.add_argument('--layers-out', default=None, help='write the per-launch profile (layer, kernel, ms, TFLOP/s, GB/s) to this file')
    return ap.parse_args()


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC passes (profiles/*pmc_summary.txt):
    2 x FETCH_SIZE (gfx950 counts wide coalesced reads at half size, MI355X_MICROARCH.md) + WRITE_SIZE.  None if absent."""
    import glob
    key = kernel_name.replace(' ', '')
    for path in sorted(glob.glob(str(ROOT / 'profiles' / '*pmc_summary.txt')), reverse=True):
        for line in open(path):
            if line.startswith('#') or line.startswith('kernel'):
                continue
            cols = line.split()
            if cols and cols[0] == key:                     # the summary writes kernel names as one token without spaces
                try:
                    return (2.0 * float(cols[-2]) + float(cols[-1])) * 1024.0, Path(path).name
                except (ValueError, IndexError):
                    return None, None
    return None, None


def main():
    args = parse()
    import torch
    from realtime_yukarin_amd import engine, synth
    from realtime_yukarin_amd.netspec import flops as net_flops, pad_frames

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d ... bench.py --gpus %d' % (args.gpus, args.gpus))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: no HIP device visible (there is no CPU path)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    N, Wn = args.frames, args.windows
    T = N + pad_frames(N)
    d1, d2 = synth.model_descs(args.model)
    # ---- weights: rank 0 builds the blobs, one RCCL broadcast each over xGMI, every rank adopts the device buffer
    from realtime_yukarin_amd import dist as rdist
    from realtime_yukarin_amd.weights import synthetic_params
    ctx = engine.get_context(local_rank)
    P1 = synthetic_params(d1, synth.SEED_STAGE1) if rank == 0 else None
    P2 = synthetic_params(d2, synth.SEED_STAGE2) if rank == 0 else None
    net1 = rdist.make_net(ctx, d1, rdist.broadcast_blob(d1, P1, dev))
    net2 = rdist.make_net(ctx, d2, rdist.broadcast_blob(d2, P2, dev), width=synth.FFT_BINS - 1)
    del P1, P2
    if args.dtype != 'f32':
        net2.set_dtype(args.dtype)

    # ---- synthetic windows, resident in HBM before the timed region (different data per rank)
    x1 = torch.from_numpy(synth.stage1_input(N, Wn, seed=synth.SEED_INPUT + 10 * rank)).to(dev)
    sp = torch.from_numpy(synth.stage2_input(N, Wn, seed=synth.SEED_INPUT + 10 * rank + 1)).to(dev)
    y1 = torch.empty(Wn, N, d1.out_ch, dtype=torch.float32, device=dev)
    y2 = torch.empty_like(sp)
    torch.cuda.synchronize()

    if args.profile_only:
        xin = torch.randn(Wn, T, synth.FFT_BINS - 1, device=dev); yout = torch.empty_like(xin)
        for _ in range(3):
            net2.forward_device(xin.data_ptr(), yout.data_ptr(), Wn, T)
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
        if os.environ.get('RY_TIMING'):
            import ctypes
            buf = (ctypes.c_ulonglong * 8)()
            ctx.lib.dll.ry_debug_igemm_phases.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
            ctx.lib.check(ctx.lib.dll.ry_debug_igemm_phases(ctx.handle, buf))
            tot = float(sum(buf[:7])) or 1.0
            names = ['barrier1', 'lds_write(+vmcnt)', 'barrier2', 'setup/loads', 'mfma steps', 'prologue', 'epilogue']
            print('   phases: ' + ' '.join('%s=%.1f%%' % (n, 100.0 * buf[i] / tot) for i, n in enumerate(names)) + '  waves=%d cyc/wave=%.0f' % (buf[7], tot / max(buf[7], 1)), flush=True)
        return

    def step():
        net1.convert_device(x1.data_ptr(), y1.data_ptr(), Wn, N)
        net2.convert_device(sp.data_ptr(), y2.data_ptr(), Wn, N)

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed():
        # W untimed warm-up steps, then exactly K steps between barrier + synchronize fences; the maximum over the ranks
        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        ctx.timer_start()
        for _ in range(args.steps):
            step()
        dms = ctx.timer_stop()
        fence()
        el = time.perf_counter() - t0
        if dist is not None:
            te = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            el = float(te.item())
        assert bool(torch.isfinite(y2).all()) and bool(torch.isfinite(y1).all())
        return el, dms

    elapsed, dev_ms = timed()

    def time_only(fn, reps=20):
        for _ in range(3):
            fn()
        ctx.sync(); ctx.timer_start()
        for _ in range(reps):
            fn()
        return ctx.timer_stop() / reps
    # PCIe-inclusive single-window latency of the drop-in call path (host arrays in, host arrays out; never the headline value)
    host_ms = None
    if rank == 0:
        from realtime_yukarin_amd import sptk
        core = engine.VcCore(net1, net2, sptk.mc2sp_matrix(d1.out_ch - 1, sptk.mcepalpha(16000), 2 * (synth.FFT_BINS - 1)))
        xh = synth.stage1_input(N)[0]; eff = numpy.ones(N, bool)
        for _ in range(3):
            core.convert(xh, eff)
        th = time.perf_counter()
        for _ in range(20):
            core.convert(xh, eff)
        host_ms = (time.perf_counter() - th) / 20 * 1e3
        core.close()
    s1_ms = time_only(lambda: net1.convert_device(x1.data_ptr(), y1.data_ptr(), Wn, N))
    s2_ms = time_only(lambda: net2.convert_device(sp.data_ptr(), y2.data_ptr(), Wn, N))
    # live per-launch profile (HIP events around every launch of both predictors) for the roofline objects: taken straight after
    # the timed region, before the cold-cache and split-bf16 extras below change what is resident and how warm the chip is
    st2 = st1 = None
    if rank == 0:
        st2 = net2.profile(Wn, T, args.profile_reps)
        st1 = net1.profile(Wn, T, args.profile_reps)
    # stage-1 with its filters evicted (cold next to warm): 512 MiB of scratch is rewritten before every replay,
    # which pushes the 54 MB of filters out of the L2s and the 256 MB MALL, so this replay streams them from HBM
    s1_cold_ms = None
    if rank == 0:
        scratch = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        cold = []
        for i in range(6):
            scratch.fill_(i & 1); torch.cuda.synchronize(); ctx.sync()
            ctx.timer_start()
            net1.convert_device(x1.data_ptr(), y1.data_ptr(), Wn, N)
            cold.append(ctx.timer_stop())
        s1_cold_ms = sorted(cold[1:])[len(cold[1:]) // 2]
        del scratch

    # the same K steps with stage-2 in split-bf16 mode (three bf16 products per fp32 product on the bf16 matrix pipe, fp32
    # accumulate) -- reported BESIDE the exact-fp32 headline, never as `value` of an f32 run
    split = None
    if args.dtype == 'f32' and not args.no_split_bf16:
        y2_f32 = y2.clone()
        net2.set_dtype('bf16x3')
        el3, _ = timed()
        s2x_ms = time_only(lambda: net2.convert_device(sp.data_ptr(), y2.data_ptr(), Wn, N))
        torch.cuda.synchronize(); ctx.sync()
        err = float(((y2 / y2_f32) - 1.0).abs().max().item())
        lerr = float(((y2.log() - y2_f32.log()).abs().max() / y2_f32.log().abs().max()).item())
        net2.set_dtype('f32')
        split = {'dtype': 'bf16x3', 'value': round(world * Wn * N * args.steps / el3, 1), 'unit': 'frames/s',
                 'ms_per_step': round(el3 / args.steps * 1e3, 4), 'stage2_alone_ms': round(s2x_ms, 4),
                 'max_rel_diff_vs_f32_path': err, 'log_spectrum_diff_vs_f32_path': lerr, 'parity_bar': 1e-4,
                 'note': 'stage-2 MFMA-bound layers as hi*hi + lo*hi + hi*lo bf16 products with fp32 accumulation (rank 0 output compared); '
                         'opt-in mode (--dtype bf16x3 / ry_net_set_dtype(net, 2)); the headline value above is exact fp32'}
        del y2_f32

    frames_total = world * Wn * N * args.steps
    value = frames_total / elapsed
    out = {
        'metric': 'acoustic frames/s (stage1+stage2 fwd) @16kHz/5ms',
        'value': round(value, 1), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
        'x_realtime': round(value * 0.005, 1), 'x_realtime_per_gpu': round(value * 0.005 / world, 1),
        'device_ms_per_step_rank0': round(dev_ms / args.steps, 4),
        'graph_replay_ms': {'stage1_alone': round(s1_ms, 4), 'stage2_alone': round(s2_ms, 4),
                            'stage1_alone_cold': None if s1_cold_ms is None else round(s1_cold_ms, 4)},
        'host_call_ms_per_window': None if host_ms is None else round(host_ms, 4),
        'split_bf16': split,
        'config': {'workload': 'BASELINE config #3: stage-1 + stage-2 SR forward, buffer_time 0.5 s + 2x0.5 s convert_extra_time '
                               '@16 kHz / 5 ms -> %d real frames (%d padded) per window, %d window(s) per GPU per step, %s random-init weights'
                               % (N, T, Wn, args.model),
                   'model': args.model, 'frames': N, 'padded_frames': T, 'windows_per_gpu': Wn,
                   'parallelism': 'chunk-dp%d (independent windows, RCCL weight broadcast at init, no steady-state collective)' % world},
    }

    if rank == 0:
        # ---- roofline of the dominant kernel: HIP events around every launch of the stage-2 predictor, live
        if args.layers_out:
            with open(args.layers_out, 'w') as f:
                for tag, st in (('stage1', st1), ('stage2', st2)):
                    for s in st:
                        f.write('%-7s %-12s %-26s grid=%-16s %9.2f us %8.2f TFLOP/s %9.1f GB/s\n' % (
                            tag, s['layer'], s['name'], 'x'.join(map(str, s['grid'])), s['ms'] * 1e3,
                            s['flops'] / max(s['ms'], 1e-9) / 1e9, s['bytes'] / max(s['ms'], 1e-9) / 1e6))
        fam = {}
        for s in st2 + st1:
            f = fam.setdefault(s['name'], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
            f['ms'] += s['ms']; f['flops'] += s['flops']; f['bytes'] += s['bytes']; f['launches'] += 1
        # dominant kernel = the family with the most time in the stage that bounds the step (stage 2; stage 1 overlaps on its own
        # stream and has its own HBM roofline object below)
        st2_names = set(s['name'] for s in st2)
        dom = max(((k, v) for k, v in fam.items() if k in st2_names), key=lambda kv: kv[1]['ms'])
        dname, dv = dom
        ach = dv['flops'] / (dv['ms'] * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic(dname)
        targs = dname[dname.find('<') + 1:-1].split(',') if dname.startswith('ry_igemm_ldsdma<') else []
        is_bf16 = len(targs) > 5 and targs[5] == 'true'                # ry_igemm_ldsdma<BM,BN,WM,WN,KG,BF16,PATCH>
        peak_tf = 2500.0 if is_bf16 else F32_MFMA_PEAK_TF             # dense bf16 MFMA peak, else fp32-input MFMA peak
        out['roofline'] = {'kernel': dname, 'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak_tf, 'unit': 'TFLOP/s',
                           'frac': round(ach / peak_tf, 4), 'traffic': traffic, 'traffic_unit': 'HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE)',
                           'traffic_source': traffic_src,
                           'launches': dv['launches'], 'avg_launch_ms': round(dv['ms'] / dv['launches'], 4),
                           'alg_flops_per_launch': dv['flops'] / dv['launches']}
        if args.dtype == 'bf16x3' and is_bf16:      # the matrix pipe executes three bf16 products per algorithmic (fp32) product
            out['roofline']['mfma_flops_per_alg_flop'] = 3
        c1 = [s for s in st1 if s['name'].startswith(('ry_c1d_os', 'ry_conv1d_ws'))]
        ms1 = sum(s['ms'] for s in c1); by1 = sum(s['bytes'] for s in c1)
        out['roofline_stage1'] = {'kernel': '%s<*> (%d launches)' % (c1[0]['name'].split('<')[0], len(c1)), 'bound': 'hbm', 'achieved': round(by1 / (ms1 * 1e-3) / 1e9, 1),
                                  'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(by1 / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  'traffic': None, 'alg_bytes_per_forward': by1, 'kernel_ms_per_forward': round(ms1, 4),
                                  'frac_graph_warm': round(by1 / (s1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  'frac_graph_cold': None if not s1_cold_ms else round(by1 / (s1_cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        out['kernels'] = {k: {'ms': round(v['ms'], 4), 'launches': v['launches'], 'tflops': round(v['flops'] / max(v['ms'], 1e-9) / 1e9, 2)}
                          for k, v in sorted(fam.items(), key=lambda kv: -kv[1]['ms'])}
        out['stage_ms'] = {'stage1_kernels': round(sum(s['ms'] for s in st1), 4), 'stage2_kernels': round(sum(s['ms'] for s in st2), 4)}
        out['alg'] = {'stage1_gflop': net_flops(d1, T) * Wn / 1e9, 'stage2_gflop': net_flops(d2, T, synth.FFT_BINS - 1) * Wn / 1e9}

        # ---- CPU baseline beside it: the oracle's torch/oneDNN restatement on this box's host cores (bounded sample)
        if not args.no_cpu_baseline and world == 1:
            from oracle import torch_ref
            from realtime_yukarin_amd.weights import synthetic_params
            import torch as _t
            P1 = synthetic_params(d1, synth.SEED_STAGE1); P2 = synthetic_params(d2, synth.SEED_STAGE2)
            t1n, t2n = torch_ref.TorchUNet(P1), torch_ref.TorchUNet(P2)
            xs = synth.stage1_input(N)[0]; sps = synth.stage2_input(N)[0]
            torch_ref.stage1_convert_core(t1n, xs); torch_ref.stage2_convert(t2n, sps)   # warm-up
            reps, tb = 0, time.perf_counter()
            while True:
                torch_ref.stage1_convert_core(t1n, xs); torch_ref.stage2_convert(t2n, sps)
                reps += 1
                if time.perf_counter() - tb > args.cpu_seconds or reps >= 50:
                    break
            cpu_s = (time.perf_counter() - tb) / reps
            out['cpu_baseline'] = {'value': round(N / cpu_s, 1), 'unit': 'frames/s', 'cores': _t.get_num_threads(), 'kind': 'port',
                                   'sample': '%d x (stage-1 + stage-2 convert of one %d-frame window), CPU restatement (torch/oneDNN fp32), not Chainer; host has %d logical cpus'
                                             % (reps, N, os.cpu_count())}
        print(json.dumps(out), flush=True)
    net1.close(); net2.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
