// ry_net.cpp -- the predictor C ABI of libry355.so (include/ry355.h): context and predictor lifetime, dtype modes, the forward / convert entry points,
// profiling and planner debug hooks.  Planner: ry_plan.cpp; kernels + launchers + single operators: ry_exec.cpp; shared declarations: ry_plan.h /
// ry_host.h; window call: ry_vc.cpp.
//
// Build (product): hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c <each unit>, linked into libry355.so   (realtime_yukarin_amd/build.py)
// Build (test emulator, no GPU): clang++ -x c++ -DRY_HOST_EMU ... the same units + tests/emu/ry_emu.cpp
//
// What the reference does here (all inside un-vendored dependencies, [MEM]): Chainer builds the
// predictor from config.model, load_npz fills it, to_gpu moves it, and every convert() call runs 16
// conv/deconv links + BatchNormalization + activations one cuDNN/CuPy launch at a time with a
// materialised F.concat per decoder layer (call sites: realtime_voice_conversion/yukarin_wrapper/
// voice_changer.py:33,41).  Here: filters are re-laid out once at creation (BN folded to scale/shift),
// a plan per (batch, frames) owns all activation buffers, and the whole forward -- pad/log wrapper
// kernels, 16 fused layers, exp/crop -- is captured once into a hipGraph and replayed per buffer.
#include "ry_plan.h"

thread_local std::string g_ry_err;

extern "C" {


const char* ry_last_error(void) { return g_ry_err.c_str(); }

int ry_device_count(void) {
    int n = 0;
    if (rt::device_count(&n) != 0) return 0;
    return n;
}

int ry_init(int device, ry_ctx** out) {
    if (!out) return fail(RY_EINVAL, "null out pointer");
    *out = nullptr;
    int n = 0;
    RT_TRY(rt::device_count(&n));
    if (device < 0 || device >= n) return fail(RY_EINVAL, "device %d out of range (%d visible)", device, n);
    RT_TRY(rt::set_device(device));
    std::unique_ptr<ry_ctx> c(new ry_ctx());
    c->device = device;
    RT_TRY(rt::stream_create(&c->stream));
    RT_TRY(rt::event_create(&c->t0));
    RT_TRY(rt::event_create(&c->t1));
    c->timers = true;
    RY_TRY(read_env_switches());
    *out = c.release();
    return RY_OK;
}

void ry_shutdown(ry_ctx* ctx) {
    if (!ctx) return;
    rt::set_device(ctx->device);
    rt::stream_sync(ctx->stream);
    if (ctx->timers) { rt::event_destroy(ctx->t0); rt::event_destroy(ctx->t1); }
    rt::stream_destroy(ctx->stream);
    for (void* q : ctx->owned) rt::dfree(q);
    delete ctx;
}

int ry_sync(ry_ctx* ctx) {
    if (!ctx) return fail(RY_ESTATE, "null context");
    for (ry_net* n : ctx->nets) RT_TRY(rt::stream_sync(n->stream));
    RT_TRY(rt::stream_sync(ctx->stream));
    return RY_OK;
}

void* ry_stream(ry_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int ry_timer_start(ry_ctx* ctx) {
    if (!ctx) return fail(RY_ESTATE, "null context");
    // Everything queued so far is waited for on the HOST, then t0 is recorded and waited for: whatever is enqueued afterwards starts after
    // t0 without a cross-stream wait.  (The former form -- every predictor stream waits on t0 with hipStreamWaitEvent -- left the two
    // window lanes of ry_vc serialised for the rest of the run: 1.32 instead of 1.16 ms per window, round 2.)
    for (ry_net* n : ctx->nets) RT_TRY(rt::stream_sync(n->stream));
    RT_TRY(rt::stream_sync(ctx->stream));
    RT_TRY(rt::event_record(ctx->t0, ctx->stream));
    RT_TRY(rt::event_sync(ctx->t0));
    return RY_OK;
}

int ry_timer_stop(ry_ctx* ctx, float* ms) {
    if (!ctx || !ms) return fail(RY_EINVAL, "null argument");
    // t1 after every predictor stream, joined on the HOST: an event record on each predictor stream plus hipStreamWaitEvent from the
    // context stream, issued while the two window lanes of ry_vc still had work queued, cost them their overlap for the rest of the run
    // (1.33 instead of 1.16 ms per window, BENCH_TIMER_MODE sweep of round 2)
    for (ry_net* n : ctx->nets) RT_TRY(rt::stream_sync(n->stream));
    RT_TRY(rt::event_record(ctx->t1, ctx->stream));
    RT_TRY(rt::event_sync(ctx->t1));
    RT_TRY(rt::event_elapsed(ms, ctx->t0, ctx->t1));
    return RY_OK;
}

size_t ry_net_param_count(const ry_net_desc* desc) {
    if (check_desc(desc) != RY_OK) return 0;
    size_t n = 0;
    for (const Layer& l : build_topology(*desc)) n += layer_param_count(l, desc->ndim);
    return n;
}

int ry_net_create(ry_ctx* ctx, const ry_net_desc* desc, const float* weights, size_t n_floats, int on_device, ry_net** out) {
    if (!ctx || !out || !weights) return fail(RY_EINVAL, "null argument");
    *out = nullptr;
    RY_TRY(check_desc(desc));
    const size_t want = ry_net_param_count(desc);
    if (n_floats != want) return fail(RY_EINVAL, "weight blob has %zu floats, the predictor needs %zu", n_floats, want);
    RT_TRY(rt::set_device(ctx->device));
    std::vector<float> host;
    const float* blob = weights;
    if (on_device) {
        host.resize(n_floats);
        RT_TRY(rt::d2h(host.data(), weights, n_floats * sizeof(float), ctx->stream));
        RT_TRY(rt::stream_sync(ctx->stream));
        blob = host.data();
    }
    std::unique_ptr<ry_net> net(new ry_net());
    net->ctx = ctx;
    net->desc = *desc;
    if (net->desc.bn_eps <= 0.f) net->desc.bn_eps = 2e-5f;
    net->layers = build_topology(*desc);
    if (const char* e = getenv("RY_GRAPH")) net->use_graph = atoi(e) != 0;
    RT_TRY(rt::stream_create(&net->stream));
    RT_TRY(rt::event_create_fast(&net->done));
    net->has_done = true;
    size_t off = 0;
    for (Layer& l : net->layers) {
        const size_t nw = (size_t)l.cin() * l.cout * ipow((size_t)l.k, desc->ndim);
        const float* W = blob + off; off += nw;
        const float* b = blob + off; off += l.cout;
        const float* bn = nullptr;
        if (l.bn) { bn = blob + off; off += 4 * (size_t)l.cout; }
        RY_TRY(prepare_layer(ctx, *net->weights, l, desc->ndim, net->desc.bn_eps, W, b, bn));
    }
    ctx->nets.push_back(net.get());
    *out = net.release();
    return RY_OK;
}

// A second handle on the same predictor: the filters stay where they are (one copy in HBM, freed with the last handle), the clone
// gets its own stream, launch plans, activation buffers and captured graphs -- what a window needs to run beside another one.
int ry_net_clone(ry_net* src, ry_net** out) {
    if (!src || !out) return fail(RY_EINVAL, "null argument");
    *out = nullptr;
    ry_ctx* ctx = src->ctx;
    RT_TRY(rt::set_device(ctx->device));
    std::unique_ptr<ry_net> net(new ry_net());
    net->ctx = ctx; net->desc = src->desc; net->dtype = src->dtype; net->use_graph = src->use_graph;
    net->layers = src->layers;                       // device pointers into the shared arena
    net->weights = src->weights;
    RT_TRY(rt::stream_create(&net->stream));
    RT_TRY(rt::event_create_fast(&net->done));
    net->has_done = true;
    ctx->nets.push_back(net.get());
    *out = net.release();
    return RY_OK;
}

void ry_net_destroy(ry_net* net) {
    if (!net) return;
    rt::set_device(net->ctx->device);
    rt::stream_sync(net->stream);
    rt::stream_sync(net->ctx->stream);
    auto& v = net->ctx->nets;
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i] == net) { v.erase(v.begin() + i); break; }
    net->plans.clear();
    if (net->has_done) rt::event_destroy(net->done);
    rt::stream_destroy(net->stream);
    delete net;
}

int ry_net_set_dtype(ry_net* net, int dtype) {
    if (!net) return fail(RY_EINVAL, "null argument");
    if (dtype < 0 || dtype > 2) return fail(RY_EINVAL, "dtype must be 0 (fp32), 1 (bf16 operands, fp32 accumulate) or 2 (split-bf16: three bf16 products per fp32 product, fp32 accumulate)");
    if (dtype != 0 && net->desc.ndim != 2) return fail(RY_EINVAL, "the bf16 variants exist for the stage-2 predictor only");
    ry_ctx* ctx = net->ctx;
    RT_TRY(rt::set_device(ctx->device));
    RT_TRY(rt::stream_sync(net->stream));
    if (dtype == 1) {
        for (Layer& l : net->layers) {
            if (!l.wig || l.wig16 || l.cin_a % 64 != 0 || l.cin_b % 64 != 0) continue;
            const TapTable t = make_taps(l);
            const int C = l.cin(), N = l.cout;
            const size_t n = (size_t)t.nphases * N * t.ntaps * C;
            std::vector<float> w32(n);
            RT_TRY(rt::d2h(w32.data(), l.wig, n * sizeof(float), ctx->stream));
            RT_TRY(rt::stream_sync(ctx->stream));
            // fp32 blocks [..][C/32][64][32]  ->  bf16 blocks [..][C/64][64][64]
            std::vector<unsigned short> w16(n);
            const size_t outer = (size_t)t.nphases * (N / 64) * t.ntaps;
            for (size_t o = 0; o < outer; ++o)
                for (int c = 0; c < C; ++c)
                    for (int nl = 0; nl < 64; ++nl)
                        w16[(o * (C / 64) + c / 64) * 4096 + wig16_inblock(nl, c % 64)] = host_f2bf(w32[(o * (C / 32) + c / 32) * 2048 + wig_inblock(nl, c % 32)]);
            float* d = nullptr;
            RY_TRY(net->weights->alloc(&d, (n + 1) / 2));
            RT_TRY(rt::h2d(d, w16.data(), n * sizeof(unsigned short), ctx->stream));
            RT_TRY(rt::stream_sync(ctx->stream));
            l.wig16 = d;
        }
    }
    RY_TRY(read_plan_env());
    if (dtype == 2) {
        if (const char* e = getenv("RY_X3_MINM")) g_x3_min_m = atoi(e);      // read again here so that a test / sweep can move it per call
        for (Layer& l : net->layers) {
            if (!l.wig || l.wigx3 || l.cin_a % 64 != 0 || l.cin_b % 64 != 0) continue;
            const TapTable t = make_taps(l);
            const size_t n = (size_t)t.nphases * l.cout * t.ntaps * l.cin();
            std::vector<float> w32(n);
            RT_TRY(rt::d2h(w32.data(), l.wig, n * sizeof(float), ctx->stream));
            RT_TRY(rt::stream_sync(ctx->stream));
            std::vector<unsigned short> wx;
            build_wigx3(l, w32, wx);
            float* d = nullptr;
            RY_TRY(net->weights->alloc(&d, (wx.size() + 1) / 2));
            RT_TRY(rt::h2d(d, wx.data(), wx.size() * sizeof(unsigned short), ctx->stream));
            RT_TRY(rt::stream_sync(ctx->stream));
            l.wigx3 = d;
        }
    }
    net->dtype = dtype;
    net->plans.clear();                              // launch plans (and captured graphs) depend on the kernel choice
    return RY_OK;
}

int ry_net_forward(ry_net* net, const float* x, float* y, int batch, int frames, int on_device) {
    if (!net || !x || !y) return fail(RY_EINVAL, "null argument");
    Plan* P = nullptr;
    RY_TRY(get_plan(net, batch, frames, 0, 0, &P));
    return run_plan(net, *P, x, y, on_device);
}

static int convert_common(ry_net* net, int want_ndim, const float* x, float* y, int batch, int n_frames, int on_device,
                          int disc_front = 0, int disc_back = 0) {
    if (!net || !x || !y) return fail(RY_EINVAL, "null argument");
    if (net->desc.ndim != want_ndim) return fail(RY_EINVAL, "wrong predictor: this call needs a stage-%d net", want_ndim);
    if (n_frames < 1) return fail(RY_EINVAL, "n_frames must be positive (got %d)", n_frames);
    if (disc_front < 0 || disc_back < 0) return fail(RY_EINVAL, "discarded frame counts must not be negative (got %d, %d)", disc_front, disc_back);
    if (disc_front >= (1 << 20) || disc_back >= (1 << 20)) return fail(RY_EINVAL, "discarded frame counts are limited to 2^20");
    const int T = n_frames + (128 - n_frames % 128);
    Plan* P = nullptr;
    RY_TRY(get_plan(net, batch, T, 1, n_frames, &P));
    P->disc_front = disc_front; P->disc_back = disc_back;
    RY_TRY(run_plan(net, *P, x, y, on_device));
    if (!on_device && want_ndim == 2 && (disc_front > 0 || disc_back > 0)) {       // host arrays: the rows that were not computed come back as zeros
        const int k0 = disc_front < n_frames ? disc_front : 0;
        const int k1 = n_frames - disc_back > k0 ? n_frames - disc_back : n_frames;
        const size_t cols = (size_t)net->desc.width + 1;
        for (int b = 0; b < batch; ++b) {
            float* yb = y + (size_t)b * n_frames * cols;
            if (k0 > 0) memset(yb, 0, (size_t)k0 * cols * sizeof(float));
            if (k1 < n_frames) memset(yb + (size_t)k1 * cols, 0, (size_t)(n_frames - k1) * cols * sizeof(float));
        }
    }
    return RY_OK;
}

int ry_ac_convert(ry_net* net, const float* x, float* y, int batch, int n_frames, int on_device) {
    return convert_common(net, 1, x, y, batch, n_frames, on_device);
}

int ry_sr_convert(ry_net* net, const float* sp, float* out, int batch, int n_frames, int on_device) {
    return convert_common(net, 2, sp, out, batch, n_frames, on_device);
}

// The same for a caller that will throw away the first `discard_front` and the last `discard_back` frames of every window (the live
// caller: ConvertStream.process picks [pad, -pad) of what it converted, realtime_voice_conversion/stream/convert_stream.py:40-42):
// rows [discard_front, n_frames - discard_back) of `out` are written, bit-identical to ry_sr_convert; the other rows are left untouched.
int ry_sr_convert_rows(ry_net* net, const float* sp, float* out, int batch, int n_frames, int discard_front, int discard_back, int on_device) {
    return convert_common(net, 2, sp, out, batch, n_frames, on_device, discard_front, discard_back);
}

int ry_net_profile(ry_net* net, int batch, int frames, int reps, ry_kernel_stat* stats, int max_stats, int* n_stats) {
    if (!net || !stats || !n_stats || reps < 1) return fail(RY_EINVAL, "bad argument");
    Plan* P = nullptr;
    RY_TRY(get_plan(net, batch, frames, 0, 0, &P));
    return profile_plan(net, P, reps, stats, max_stats, n_stats);
}

int ry_net_profile_window(ry_net* net, int n_frames, int reps, ry_kernel_stat* stats, int max_stats, int* n_stats) {
    if (!net || !stats || !n_stats || reps < 1 || n_frames < 1) return fail(RY_EINVAL, "bad argument");
    Plan* P = nullptr;
    RY_TRY(get_plan(net, 1, n_frames + (128 - n_frames % 128), 1, n_frames, &P));
    P->disc_front = P->disc_back = 0;
    return profile_plan(net, P, reps, stats, max_stats, n_stats);
}

// diagnostics / tests: read the process-wide RY_* switches again (they are otherwise read when a context is created; launch plans built before keep
// their choices until ry_net_set_dtype drops them)
int ry_debug_reload_env(void) { return read_env_switches(); }

// diagnostics: the plan (tile, external splits, K groups, estimated time) the stage-2 planner picks for one layer shape
int ry_debug_plan_igemm(int M, int Cout, int nphases, int nk, int* tile, int* splits, int* kgroups, double* est_us) {
    if (!tile || !splits || !kgroups) return fail(RY_EINVAL, "null argument");
    if (M < 1 || Cout < 64 || Cout % 64 != 0 || nphases < 1 || nk < 1) return fail(RY_EINVAL, "not an implicit-GEMM layer shape");
    Layer l; l.cout = Cout;
    *tile = 0; *splits = 0; *kgroups = 0;
    choose_igemm(l, M, nphases, nk, tile, splits, kgroups);
    if (est_us) {
        int bm, bn; tile_dims(*tile, &bm, &bn);
        *est_us = est_time((long)((M + bm - 1) / bm) * (Cout / bn) * nphases, bm, bn, *splits, tile_occ(*tile, *kgroups), *kgroups, M, Cout, nk);
    }
    return RY_OK;
}

// the same for the bf16 (mode 1) / split-bf16 (mode 2) kernels (nk counts 64-channel chunks: taps x Cin / 64, three times that in
// split-bf16 mode); non-zero *tile / *splits / *kgroups on entry are kept (the estimate of a given plan)
int ry_debug_plan_igemm_bf16(int mode, int M, int Cout, int nphases, int nk, int* tile, int* splits, int* kgroups, double* est_us) {
    if (mode != 1 && mode != 2) return fail(RY_EINVAL, "mode must be 1 (bf16) or 2 (split-bf16)");
    if (!tile || !splits || !kgroups) return fail(RY_EINVAL, "null argument");
    if (M < 1 || Cout < 64 || Cout % 64 != 0 || nphases < 1 || nk < 1) return fail(RY_EINVAL, "not an implicit-GEMM layer shape");
    Layer l; l.cout = Cout;
    choose_igemm(l, M, nphases, nk, tile, splits, kgroups, mode);
    if (est_us) {
        int bm, bn; tile_dims(*tile, &bm, &bn);
        *est_us = est_time((long)((M + bm - 1) / bm) * (Cout / bn) * nphases, bm, bn, *splits, tile_occ(*tile, *kgroups), *kgroups, M, Cout, nk)
                  * (M > 64 && Cout % 128 == 0 ? (1.0 / bm + 1.0 / bn) * 64.0 : 1.0);
    }
    return RY_OK;
}

// diagnostics: the output-stationary slice the planner picks for one layer shape (choose_os2)
int ry_debug_plan_os2(int M, int Cout, int nphases, int units, int* mt4, int* nt4, int* waves, int* depth, double* cost) {
    if (!mt4 || !nt4 || !waves || !depth) return fail(RY_EINVAL, "null argument");
    if (M < 1 || Cout < 4 || Cout % 4 != 0 || nphases < 1 || units < 1) return fail(RY_EINVAL, "not an output-stationary layer shape");
    if (!choose_os2(M, Cout, nphases, units, mt4, nt4, waves, depth, cost))
        return fail(RY_EINVAL, "no output-stationary slice for %d rows x %d channels x %d units", M, Cout, units);
    return RY_OK;
}

int ry_debug_plan_wino(int Mh, int Mw, int Cout, int nphases, int npatches, int batch, int* cfg, int* mbw, int* splits) {
    if (!cfg || !mbw || !splits) return fail(RY_EINVAL, "null argument");
    if (Mh < 1 || Mw < 1 || Cout < 64 || Cout % 64 || (nphases != 1 && nphases != 4) || npatches < 1 || batch < 1) return fail(RY_EINVAL, "not a Winograd layer shape");
    if (!choose_wino(Mh, Mw, Cout, nphases, npatches, batch, cfg, mbw, splits)) return fail(RY_EINVAL, "no Winograd tile divides a %d x %d grid", Mh, Mw);
    return RY_OK;
}

}  // extern "C"
