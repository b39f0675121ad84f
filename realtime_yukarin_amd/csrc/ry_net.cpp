// ry_net.cpp -- predictor executor of libry355.so (C ABI: include/ry355.h): topology, filter re-layout, stage-2 planner, launch plans + graphs,
// the predictor API and the single operators.  Shared host structures: ry_host.h; window call: ry_vc.cpp; RCCL: ry_comm.cpp.
//
// Build (product): hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -x hip ry_net.cpp ry_vc.cpp ry_comm.cpp -o libry355.so   (realtime_yukarin_amd/build.py)
// Build (test emulator, no GPU): clang++ -x c++ -DRY_HOST_EMU ... the same units + tests/emu/ry_emu.cpp
//
// What the reference does here (all inside un-vendored dependencies, [MEM]): Chainer builds the
// predictor from config.model, load_npz fills it, to_gpu moves it, and every convert() call runs 16
// conv/deconv links + BatchNormalization + activations one cuDNN/CuPy launch at a time with a
// materialised F.concat per decoder layer (call sites: realtime_voice_conversion/yukarin_wrapper/
// voice_changer.py:33,41).  Here: filters are re-laid out once at creation (BN folded to scale/shift),
// a plan per (batch, frames) owns all activation buffers, and the whole forward -- pad/log wrapper
// kernels, 16 fused layers, exp/crop -- is captured once into a hipGraph and replayed per buffer.
#include "ry_kernels.h"
#include "ry_host.h"

thread_local std::string g_ry_err;

// ------------------------------------------------------------------------------------------------
// topology (same K-list order as realtime_yukarin_amd/netspec.py)
// ------------------------------------------------------------------------------------------------
static const int ENC_CH[8] = {1, 2, 4, 8, 8, 8, 8, 8};
static const int DEC_IN[7] = {8, 16, 16, 16, 16, 8, 4};
static const int DEC_OUT[7] = {8, 8, 8, 8, 4, 2, 1};

static std::vector<Layer> build_topology(const ry_net_desc& d) {
    std::vector<Layer> L(16);
    const int B = d.base, e = d.extensive_layers;
    const int end_k = e > 0 ? 3 : 1;
    // glu_generator (stage 1, UNVERIFIED [MEM]): a conv + BN block computes 2 x co channels (value | gate), its consumers see co = value * sigmoid(gate)
    const int g = (d.glu && d.ndim == 1) ? 2 : 1;
    const int blk_act_e = g == 2 ? RY_ACT_GLU : RY_ACT_LRELU, blk_act_d = g == 2 ? RY_ACT_GLU : RY_ACT_RELU;
    auto nm = [](Layer& l, const char* p, int i) { snprintf(l.name, sizeof l.name, "%s/c%d", p, i); };
    {   // encoder c0: conv + bias, leaky_relu
        Layer& l = L[0]; nm(l, "encoder", 0);
        l.k = end_k; l.stride = 1; l.pad = end_k / 2; l.cin_a = d.in_ch; l.cout = B; l.src_a = -1; l.act = RY_ACT_LRELU;
    }
    for (int i = 1; i < 8; ++i) {
        Layer& l = L[i]; nm(l, "encoder", i);
        const bool down = i < e;
        l.k = down ? 4 : 1; l.stride = down ? 2 : 1; l.pad = down ? 1 : 0;
        l.cin_a = ENC_CH[i - 1] * B; l.cout = ENC_CH[i] * B * g; l.src_a = i - 1; l.bn = true; l.act = blk_act_e;
    }
    for (int j = 0; j < 7; ++j) {
        Layer& l = L[8 + j]; nm(l, "decoder", j);
        const bool up = (7 - j) < e;
        l.deconv = up; l.k = up ? 4 : 1; l.stride = up ? 2 : 1; l.pad = up ? 1 : 0;
        l.cout = DEC_OUT[j] * B * g; l.bn = true; l.act = blk_act_d;
        if (j == 0) { l.cin_a = DEC_IN[0] * B; l.src_a = 7; }
        else { l.cin_a = DEC_OUT[j - 1] * B; l.cin_b = ENC_CH[7 - j] * B; l.src_a = 8 + j - 1; l.src_b = 7 - j; }
    }
    {   // decoder c7: conv + bias on concat(decoder c6, encoder c0)
        Layer& l = L[15]; nm(l, "decoder", 7);
        l.k = end_k; l.stride = 1; l.pad = end_k / 2; l.cin_a = B; l.cin_b = B; l.cout = d.out_ch; l.src_a = 14; l.src_b = 0;
        l.act = RY_ACT_NONE;
    }
    return L;
}

static size_t ipow(size_t b, int e) { size_t r = 1; while (e-- > 0) r *= b; return r; }

static size_t layer_param_count(const Layer& l, int ndim) {
    size_t n = (size_t)l.cin() * l.cout * ipow((size_t)l.k, ndim) + l.cout;
    if (l.bn) n += 4 * (size_t)l.cout;
    return n;
}

static int check_desc(const ry_net_desc* d) {
    if (!d) return fail(RY_EINVAL, "null descriptor");
    if (d->ndim != 1 && d->ndim != 2) return fail(RY_EINVAL, "ndim must be 1 or 2 (got %d)", d->ndim);
    if (d->in_ch < 1 || d->out_ch < 1 || d->base < 1) return fail(RY_EINVAL, "in_ch/out_ch/base must be positive");
    if (d->extensive_layers < 0 || d->extensive_layers > 8) return fail(RY_EINVAL, "extensive_layers must be in 0..8");
    if (d->ndim == 2 && d->width < 1) return fail(RY_EINVAL, "stage-2 needs width >= 1");
    if (d->ndim == 2 && (d->in_ch != 1 || d->out_ch != 1))
        return fail(RY_EINVAL, "stage-2 (SRPredictor) takes and returns one channel (got %d -> %d)", d->in_ch, d->out_ch);
    if (d->glu != 0 && d->glu != 1) return fail(RY_EINVAL, "glu must be 0 or 1 (got %d)", d->glu);
    if (d->glu && d->ndim != 1) return fail(RY_EINVAL, "glu_generator is a stage-1 option");
    return RY_OK;
}

// ------------------------------------------------------------------------------------------------
// host-side filter re-layout and BatchNormalization folding (once, at creation)
// ------------------------------------------------------------------------------------------------
static const int DECONV_KY[2][2] = {{1, 3}, {0, 2}};   // output parity p, tap t -> kernel index
static const int DECONV_DY[2][2] = {{0, -1}, {1, 0}};  // output parity p, tap t -> input offset

static void fold_scale_shift(const Layer& l, const float* b, const float* bn, float eps,
                             std::vector<float>& scale, std::vector<float>& shift) {
    scale.resize(l.cout); shift.resize(l.cout);
    for (int c = 0; c < l.cout; ++c) {
        if (bn) {
            const double g = bn[c], be = bn[l.cout + c], mu = bn[2 * l.cout + c], var = bn[3 * l.cout + c];
            const double s = g / std::sqrt(var + (double)eps);
            scale[c] = (float)s;
            shift[c] = (float)(((double)(b ? b[c] : 0.f) - mu) * s + be);
        } else {
            scale[c] = 1.f;
            shift[c] = b ? b[c] : 0.f;
        }
    }
}

// stage-1: [Ctot][N][4], taps beyond k zero.  conv W (N, C, k); deconv W (C, N, 4)
static void relayout_1d(const Layer& l, const float* W, std::vector<float>& out) {
    const int C = l.cin(), N = l.cout, K = l.k;
    out.assign((size_t)C * N * 4, 0.f);
    for (int c = 0; c < C; ++c)
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < K; ++k)
                out[((size_t)c * N + n) * 4 + k] = l.deconv ? W[((size_t)c * N + n) * K + k] : W[((size_t)n * C + c) * K + k];
}

// the same taps as [N][Ctot][4] for ry_c1d_os (a lane walks the input channels of one output channel: 16 bytes per lane, coalesced)
static void relayout_1d_os(const Layer& l, const float* W, std::vector<float>& out) {
    const int C = l.cin(), N = l.cout, K = l.k;
    out.assign((size_t)C * N * 4, 0.f);
    for (int c = 0; c < C; ++c)
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < K; ++k)
                out[((size_t)n * C + c) * 4 + k] = l.deconv ? W[((size_t)c * N + n) * K + k] : W[((size_t)n * C + c) * K + k];
}

struct TapTable {
    int nphases = 1, ntaps = 1;
    int dy[4][16], dx[4][16], ky[4][16], kx[4][16], pdy[4], pdx[4];
};

static TapTable make_taps(const Layer& l) {
    TapTable t;
    memset(&t, 0, sizeof t);
    if (l.deconv) {
        t.nphases = 4; t.ntaps = 4;
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                const int ph = py * 2 + px;
                t.pdy[ph] = py; t.pdx[ph] = px;
                for (int ty = 0; ty < 2; ++ty)
                    for (int tx = 0; tx < 2; ++tx) {
                        const int tt = ty * 2 + tx;
                        t.dy[ph][tt] = DECONV_DY[py][ty]; t.dx[ph][tt] = DECONV_DY[px][tx];
                        t.ky[ph][tt] = DECONV_KY[py][ty]; t.kx[ph][tt] = DECONV_KY[px][tx];
                    }
            }
    } else {
        t.nphases = 1; t.ntaps = l.k * l.k;
        for (int ky = 0; ky < l.k; ++ky)
            for (int kx = 0; kx < l.k; ++kx) {
                const int tt = ky * l.k + kx;
                t.dy[0][tt] = ky * l.dil; t.dx[0][tt] = kx * l.dil; t.ky[0][tt] = ky; t.kx[0][tt] = kx;
            }
    }
    return t;
}

// stage-2 filters: conv W (N, C, k, k); deconv W (C, N, 4, 4)
static float w2d_at(const Layer& l, const float* W, int n, int c, int ky, int kx) {
    const int C = l.cin(), N = l.cout, K = l.k;
    return l.deconv ? W[(((size_t)c * N + n) * K + ky) * K + kx] : W[(((size_t)n * C + c) * K + ky) * K + kx];
}

// implicit-GEMM filters: [phase][N/64][tap][C/32][64 couts][32 k].  Each (64 x 32) chunk a workgroup stages per K step is
// one contiguous 8 KB block: a wave's 16-byte lane loads cover 1 KB of consecutive addresses, and the rows of a B tile are
// not spread at a power-of-two stride of 4-16 KB (which funnels every workgroup's B traffic into the same L2 channels).
// Stage-2 implicit-GEMM weights: blocks [phase][N/64][tap][Ctot/32] of 64 output channels x 32 input channels, each block
// stored in MFMA FRAGMENT ORDER [n/32 : 2][s : 4][lane : 64][t : 4] with lane = 32 * lh + (n % 32) and k = 8 s + 4 lh + t:
// the 16 bytes lane `lane` feeds to the four v_mfma_f32_32x32x2_f32 of K step s are contiguous, one (n/32, s) piece is
// 1 KiB in lane order -- a wave loads its B fragments straight into registers (or a piece into LDS) fully coalesced.
static inline size_t wig_inblock(int nl, int k) {
    return (size_t)(nl >> 5) * 1024 + (size_t)(k >> 3) * 256 + (size_t)((((k >> 2) & 1) * 32 + (nl & 31)) * 4) + (size_t)(k & 3);
}

// bf16 blocks of 64 output channels x 64 input channels, same idea: [n/32 : 2][s : 4][lane : 64][j : 8] with
// lane = 32 * lh + (n % 32) and k = 16 s + 8 lh + j (the 8 bf16 a lane feeds to one v_mfma_f32_32x32x16_bf16); index in bf16 units.
static inline size_t wig16_inblock(int nl, int k) {
    return (size_t)(nl >> 5) * 2048 + (size_t)(k >> 4) * 512 + (size_t)((((k >> 3) & 1) * 32 + (nl & 31)) * 8) + (size_t)(k & 7);
}

static void relayout_igemm(const Layer& l, const float* W, std::vector<float>& out) {
    const TapTable t = make_taps(l);
    const int C = l.cin(), N = l.cout, cpt = C / 32;
    out.resize((size_t)t.nphases * N * t.ntaps * C);
    for (int ph = 0; ph < t.nphases; ++ph)
        for (int n = 0; n < N; ++n)
            for (int tt = 0; tt < t.ntaps; ++tt)
                for (int c = 0; c < C; ++c) {
                    const size_t blk = (((size_t)ph * (N / 64) + n / 64) * t.ntaps + tt) * cpt + c / 32;
                    out[blk * 2048 + wig_inblock(n % 64, c % 32)] = w2d_at(l, W, n, c, t.ky[ph][tt], t.kx[ph][tt]);
                }
}

static void relayout_direct(const Layer& l, const float* W, std::vector<float>& out) {
    const TapTable t = make_taps(l);
    const int C = l.cin(), N = l.cout;
    out.resize((size_t)t.nphases * t.ntaps * C * N);
    for (int ph = 0; ph < t.nphases; ++ph)
        for (int tt = 0; tt < t.ntaps; ++tt)
            for (int c = 0; c < C; ++c)
                for (int n = 0; n < N; ++n)
                    out[(((size_t)ph * t.ntaps + tt) * C + c) * N + n] = w2d_at(l, W, n, c, t.ky[ph][tt], t.kx[ph][tt]);
}

static bool igemm_eligible(const Layer& l) {
    return l.cin_a % 32 == 0 && l.cin_b % 32 == 0 && l.cout % 64 == 0 && l.cin() > 0;
}

// ry_c2d_os filters: [phase][N/4][tap][C/64] blocks of 4 output channels x 64 input channels, one KiB each in the order the lanes load it:
// lane = 4 * ((c % 64) / 4) + n % 4 holds the four consecutive input channels c % 4 = 0..3 of its output channel -- K position
// 4 * (lane / 4) + t of the block for the t-th v_mfma_f32_4x4x1_16B_f32 of the unit.  Consecutive (tap, chunk) units of one channel
// group are consecutive KiB: a wave streams its run of the K axis as one contiguous range.
static bool c2d_os_eligible(const Layer& l) {
    return l.cin_a % 256 == 0 && l.cin_b % 256 == 0 && l.cout % 4 == 0 && l.cin() > 0 && l.cin_a <= 2048 && l.cin_b <= 2048;    // (rounds of four 64-channel units inside one source; a source's zero pixel is ZTAIL floats)
}

static void relayout_c2d_os(const Layer& l, const float* W, std::vector<float>& out) {
    const TapTable t = make_taps(l);
    const int C = l.cin(), N = l.cout, cpt = C / 64;
    out.resize((size_t)t.nphases * N * t.ntaps * C);
    for (int ph = 0; ph < t.nphases; ++ph)
        for (int n = 0; n < N; ++n)
            for (int tt = 0; tt < t.ntaps; ++tt)
                for (int c = 0; c < C; ++c) {
                    const size_t blk = (((size_t)ph * (N / 4) + n / 4) * t.ntaps + tt) * cpt + c / 64;
                    out[blk * 256 + (size_t)((((c % 64) >> 2) * 4 + (n & 3)) * 4 + (c & 3))] = w2d_at(l, W, n, c, t.ky[ph][tt], t.kx[ph][tt]);
                }
}

// ry_wino_ldsdma filters: a k4 s2 p1 layer is a sum of 2 x 2-tap stride-1 stencils -- one per sub-pixel phase of a transposed convolution
// (taps g[a][b] = W[KY[pdy][1 - a]][KY[pdx][1 - b]] on input offset (pdy - 1 + a, pdx - 1 + b)), one per input parity (r, c) of a convolution
// (g[a][b] = W[2 a + r][2 b + c] on the parity plane) -- and each stencil's F(2x2, 2x2) filters are U = G g G^T, G = [[1, 0], [1, 1], [0, 1]],
// computed in float64 and rounded once.  Layout [phase][N / 64][slice][position i * 3 + j][n / 32 : 2][lane = 32 * lh + n % 32][t : 4] with channel
// 8 * slice' + 4 * lh + t: one (slice, position, 32 channels) piece is the KiB a wave-instruction of the kernel copies, the pieces of a slice and
// consecutive slices are consecutive.  Slices follow the kernel's K loop: deconvolution slice = channel / 8; convolution
// slice = ((channel / 16) * 4 + parity) * 2 + (channel / 8) % 2.  2.25 x the floats of the direct layout (9 positions for 4 taps).
static bool wino_eligible(const Layer& l, int ndim) {
    return ndim == 2 && l.k == 4 && l.stride == 2 && l.pad == 1 && l.dil == 1 && l.cin_a % 16 == 0 && l.cin_b % 16 == 0 && l.cout % 64 == 0 &&
           l.cin_a > 0 && l.cin_a <= 2032 && l.cin_b <= 2032;     // (the channel offset of a patch rides on the base of its zero-tail fetches: ZTAIL floats)
}

// w(n, c, ky, kx) = the layer's filter element (any accessor: the Chainer blob, or the device's direct layout read back)
template <class F>
static void relayout_wino(const Layer& l, F w, std::vector<float>& out) {
    const int C = l.cin(), N = l.cout;
    const int nph = l.deconv ? 4 : 1, nsl = l.deconv ? C / 8 : (C / 16) * 8;
    out.assign((size_t)nph * N * nsl * 72, 0.f);
    static const double G[3][2] = {{1, 0}, {1, 1}, {0, 1}};
    for (int ph = 0; ph < nph; ++ph)
        for (int n = 0; n < N; ++n)
            for (int ks = 0; ks < nsl; ++ks)
                for (int cc = 0; cc < 8; ++cc) {
                    int c; double g[2][2];
                    if (l.deconv) {
                        c = ks * 8 + cc;
                        for (int a = 0; a < 2; ++a)
                            for (int b = 0; b < 2; ++b) g[a][b] = w(n, c, DECONV_KY[ph >> 1][1 - a], DECONV_KY[ph & 1][1 - b]);
                    } else {
                        const int par = (ks >> 1) & 3;
                        c = (ks >> 3) * 16 + (ks & 1) * 8 + cc;
                        for (int a = 0; a < 2; ++a)
                            for (int b = 0; b < 2; ++b) g[a][b] = w(n, c, 2 * a + (par >> 1), 2 * b + (par & 1));
                    }
                    for (int i = 0; i < 3; ++i)
                        for (int j = 0; j < 3; ++j) {
                            double u = 0.0;
                            for (int a = 0; a < 2; ++a)
                                for (int b = 0; b < 2; ++b) u += G[i][a] * g[a][b] * G[j][b];
                            const size_t piece = ((((size_t)ph * (N / 64) + n / 64) * nsl + ks) * 9 + (i * 3 + j)) * 2 + (n % 64) / 32;
                            out[piece * 256 + (size_t)((32 * (cc >> 2) + n % 32) * 4 + (cc & 3))] = (float)u;
                        }
                }
}

// ------------------------------------------------------------------------------------------------
// per-layer launch plans
// ------------------------------------------------------------------------------------------------
// Filters above this size (floats) get the ry_c2d_os layout next to the implicit-GEMM one when a predictor is created: the layers whose time
// is the stream of their filters (SYN-64: encoder c4 .. c7, decoder c0 .. c3, 16.8 - 33.5 MB each).  Whether a plan uses it depends on the
// window (build_plan: few enough output pixels).
static size_t g_os2_min_filter = (size_t)1 << 21;      // RY_OS2_MINW (floats; tests lower it so that small predictors take the path)

static int prepare_layer(ry_ctx* ctx, Arena& arena, Layer& l, int ndim, float eps, const float* W, const float* b, const float* bn, bool want_os2 = false) {
    std::vector<float> sc, sh, w;
    fold_scale_shift(l, b, bn, eps, sc, sh);
    // scale/shift padded to a multiple of 4 floats (16-byte epilogue loads)
    sc.resize((sc.size() + 3) / 4 * 4, 1.f); sh.resize((sh.size() + 3) / 4 * 4, 0.f);
    RY_TRY(upload(arena, ctx, sc, &l.scale));
    RY_TRY(upload(arena, ctx, sh, &l.shift));
    if (ndim == 1) {
        if (l.k > 4) return fail(RY_EINVAL, "%s: 1-D kernels wider than 4 taps are not supported", l.name);
        relayout_1d(l, W, w);
        RY_TRY(upload(arena, ctx, w, &l.w1d));
        relayout_1d_os(l, W, w);
        RY_TRY(upload(arena, ctx, w, &l.w1os));
    } else {
        if (l.k * l.k > 16) return fail(RY_EINVAL, "%s: 2-D kernels larger than 4x4 are not supported", l.name);
        if (igemm_eligible(l)) { relayout_igemm(l, W, w); RY_TRY(upload(arena, ctx, w, &l.wig)); }
        if (c2d_os_eligible(l) && (want_os2 || (l.wig && (size_t)l.cin() * l.cout * l.k * l.k >= g_os2_min_filter))) {
            relayout_c2d_os(l, W, w); RY_TRY(upload(arena, ctx, w, &l.w2os));
        }
        relayout_direct(l, W, w);
        RY_TRY(upload(arena, ctx, w, &l.wdir));
    }
    return RY_OK;
}

// Activation buffers that an implicit-GEMM layer may read end in ZTAIL zeroed floats: the LDS-DMA kernel fetches its padding
// from there (RyConvGeom::zoff1 / zoff2); nothing ever writes them.
static const size_t ZTAIL = 2048;     // (round 5: a whole zeroed PIXEL of up to 2048 channels -- ry_c2d_os fetches out-of-image taps from it at any channel offset)
static int g_poison = 0;               // RY_POISON=1 (diagnostics): fresh activation buffers are filled with NaN patterns, so that a kernel that reads a row / pixel its producer
                                       // never wrote shows up as NaN in the result instead of depending on what the allocator handed out
static int alloc_ztail(ry_ctx* ctx, Arena& arena, float** p, size_t nfloats) {
    RY_TRY(arena.alloc(p, nfloats + ZTAIL));
    if (g_poison) RT_TRY(rt::dmemset(*p, 0xFF, nfloats * sizeof(float), ctx->stream));
    RT_TRY(rt::dmemset(*p + nfloats, 0, ZTAIL * sizeof(float), ctx->stream));
    RT_TRY(rt::stream_sync(ctx->stream));
    return RY_OK;
}

static void fill_geom(RyConvGeom& g, const Layer& l, const LayerPlan& lp, int B, const float* s1, int C1, const float* s2, int C2) {
    memset(&g, 0, sizeof g);
    const TapTable t = make_taps(l);
    g.src1 = s1; g.src2 = s2; g.C1 = C1; g.C2 = C2; g.S1 = C1; g.S2 = C2;
    if (lp.path == PATH_IGEMM_BF16 && lp.x3) { g.C1 = 3 * C1; g.C2 = 3 * C2; g.S1 = 2 * C1; g.S2 = 2 * C2; }   // K = [hi | lo | hi(wrapped)] over [hi | lo] pixels
    g.B = B; g.Hi = lp.Hi; g.Wi = lp.Wi; g.Ho = lp.Ho; g.Wo = lp.Wo;
    g.Hs = lp.Hi; g.Hos = lp.Ho;
    if (lp.crop_hi > 0) { g.Hi = lp.crop_hi; g.Ho = l.deconv ? 2 * lp.crop_hi : lp.crop_hi; }   // a row range of every image in the same buffers; the rows around it read as padding
    if (l.deconv) { g.Mh = g.Hi; g.Mw = lp.Wi; g.stride = 1; g.pad = 0; g.ostride = 2; }
    else { g.Mh = g.Ho; g.Mw = lp.Wo; g.stride = l.stride; g.pad = l.pad; g.ostride = 1; }
    g.nphases = t.nphases; g.ntaps = t.ntaps; g.N = l.cout; g.kw = l.deconv ? 2 : l.k; g.dil = l.deconv ? 1 : l.dil;
    const size_t esize = lp.path == PATH_IGEMM_BF16 ? 2 : 4;                  // implicit-GEMM sources end in a zeroed tail (ZTAIL floats)
    g.zoff1 = (unsigned)((size_t)B * lp.Hi * lp.Wi * g.S1 * esize); g.zoff2 = (unsigned)((size_t)B * lp.Hi * lp.Wi * g.S2 * esize);
    if (lp.crop_hi > 0 && lp.crop_lo > 0) {                                   // the range starts crop_lo rows into every image: move the bases, keep the zero tails where they are
        const size_t o1 = (size_t)lp.crop_lo * lp.Wi * g.S1 * esize, o2 = (size_t)lp.crop_lo * lp.Wi * g.S2 * esize;
        g.src1 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(s1) + o1); g.zoff1 -= (unsigned)o1;
        if (s2) { g.src2 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(s2) + o2); g.zoff2 -= (unsigned)o2; }
    }
    for (int ph = 0; ph < 4; ++ph) {
        g.pdy[ph] = (signed char)t.pdy[ph]; g.pdx[ph] = (signed char)t.pdx[ph];
        for (int tt = 0; tt < 16; ++tt) { g.tdy[ph][tt] = (signed char)t.dy[ph][tt]; g.tdx[ph][tt] = (signed char)t.dx[ph][tt]; }
    }
}

static void tile_dims(int tile, int* bm, int* bn) {
    switch (tile) {
        case TILE_128x128: *bm = 128; *bn = 128; break;
        case TILE_64x128: *bm = 64; *bn = 128; break;
        case TILE_128x64: *bm = 128; *bn = 64; break;
        case TILE_96x128: *bm = 96; *bn = 128; break;
        default: *bm = 32; *bn = 128; break;
    }
}


// Process-wide switches (INTEGRATION.md section 6 lists every one).  Round 3 removed the A/B switches of closed experiments (register-staged
// kernel, 256-row tiles, burst loads, raster tiles, two-graph cut + stagger, stage-1 tuning aids, ...): their measurements are in DESIGN.md.
static int g_s2_hole = 1;     // RY_S2_HOLE=0: the encoder computes the identical padding rows behind the real frames instead of copying them (A/B, bit-identity tests)
static int g_s2_crop = 2;     // RY_S2_CROP=0: every decoder layer of the convert wrapper computes all padded rows (A/B of the dead-row crop, used by the bit-identity tests); 1: only grids of more than one workgroup per CU
static int g_force[16][3];    // RY_PLAN="layer:tile:splits:kgroups,...": tuning aid, fixes the stage-2 plan of single layers (0 = planner's choice)
static int g_x3_min_m = 128;  // RY_X3_MINM: split-bf16 mode runs a layer on the bf16 pipe from this many GEMM rows (per phase) up (measured at 300 frames: 1 / 32 / 64 / 128 / 512 / 2048 -> 0.861 / 0.865 / 0.867 / 0.864 vs 0.840 / 0.928 ms per step on two boxes; 128 beat 512 by 1 % in the same-box A/B)
static int g_autotune = 0;    // RY_AUTOTUNE="1[:reps[:max[:pick]]]": time candidate launch plans of every stage-2 implicit-GEMM layer on the device when a plan is built (autotune_plan)
static int g_autotune_reps = 3, g_autotune_max = 0;   // ... timed rounds per candidate; cap on the candidates per layer (0 = all; tests)
static int g_autotune_pick = -1;                      // ... (tests only) take candidate `pick` of every layer instead of the fastest
static const int g_patch = 3;      // bit 0 = input-patch reuse in the deconvolution layers, bit 1 = in the k4 s2 convolution layers (DESIGN.md 5.1 + section 9: A/B measured, both on)

// Kernel names as rocprofv3 prints them (template arguments, no spaces): bench.py matches them against profiles/*.
static const char* tile_name(int tile, int kg, bool bf16, int patch) {
    {
        static std::map<int, std::string> names;        // stable storage for the returned pointers
        const int key = ((tile * 4 + kg) * 2 + (bf16 ? 1 : 0)) * 4 + patch;
        auto it = names.find(key);
        if (it == names.end()) {
            int bm, bn; tile_dims(tile, &bm, &bn);
            const int wmv = (tile == TILE_128x128) ? 2 : (tile == TILE_128x64 ? 4 : 1), wnv = 4 / wmv;
            char buf[96];
            snprintf(buf, sizeof buf, "ry_igemm_ldsdma<%d,%d,%d,%d,%d,%s,%d>", bm, bn, wmv, wnv, kg == 2 ? 2 : 1, bf16 ? "true" : "false", patch);
            it = names.emplace(key, buf).first;
        }
        return it->second.c_str();
    }
}

// ---- choice of tile, split-K and K groups for one stage-2 layer ----
// Workgroups of one tile that fit a CU.  LDS-DMA kernel: two unpadded BK = 32 buffers; register-staged kernel: one padded
// buffer, limited to 3 by its VGPR budget.
static thread_local double g_plan_peak = 157.3e6;   // flop per microsecond the planner prices the main loop at (fp32 MFMA peak; bf16: see choose_igemm)
static thread_local int g_plan_ck = 32;             // input channels per K chunk of the kernel being planned
// split-bf16 kernels (measured, profiles/r01/n_x3_plansweep_n300.txt): the main-loop rate the planner prices them at (three times
// the K of the bf16 mode per tile: the fixed costs weigh less, 128x128 tiles reach 730-800 TF of bf16 products = 0.7 x 1150),
// and the price of two K groups in one 512-thread workgroup against two 256-thread workgroups on the same CU (the GEMM alone
// ran 10-17 % slower: 69 vs 59 us on decoder c3, 73 vs 66 us on encoder c1).
static const double g_x3_peak = 2.0e9, g_x3_kg2 = 1.15;
static thread_local double g_plan_kg2 = 1.0;        // factor on the main loop of a two-K-group workgroup

static int tile_occ(int tile, int kg) {
    int bm, bn; tile_dims(tile, &bm, &bn);
    const int occ = (160 * 1024) / (kg * (bm + bn) * 32 * 4 * 2 + bm * 16);
    const int cap = 4 / kg;                                              // <= 128 VGPRs: four waves per SIMD
    return occ > cap ? cap : occ;
}

// Fraction of the MFMA peak a CU sustains with r co-resident four-wave groups running the main loop (measured on gfx950:
// a lone group cannot cover its own barriers and LDS latency).
static double cu_rate(int r) {
    static const double f[5] = {0.0, 0.36, 0.70, 0.72, 0.72};
    return f[r > 4 ? 4 : r];
}

// Estimated time (microseconds) of one layer: `blocks` output tiles of bm rows, each split over s workgroups of kg
// four-wave K groups, on 256 CUs that hold occ workgroups at a time.  The busiest CU sets the main-loop time (a partial
// last round runs at the rate of its fewer resident groups); external split-K adds the slab traffic and a reduce launch.
static double est_time(long blocks, int bm, int bn, int s, int occ, int kg, int M, int N, int nk) {
    const long g = blocks * s;
    const long per_cu = (g + 255) / 256;
    const long full = per_cu / occ, rem = per_cu % occ;
    const double tile_us = 2.0 * bm * bn * ((double)g_plan_ck * nk) / (g_plan_peak / 256.0);   // one tile on one CU at the peak (157.3 TFLOP/s over 256 CUs)
    const double w = tile_us / (double)(s * kg);                              // work of one four-wave group
    double t = (double)full * occ * kg * w / cu_rate(occ * kg) + (rem ? (double)rem * kg * w / cu_rate((int)rem * kg) : 0.0);
    if (kg > 1) t *= g_plan_kg2;
    t += 4.0 + (double)M * N * 4.0 / 5.0e6;                                     // launch + ramp, output stores at ~5 TB/s (exposed: one round)
    if (s > 1) t += 5.0 + (2.0 * s) * M * N * 4.0 / 4.0e6;                     // s slab writes + s slab reads at ~4 TB/s, reduce launch
    if (kg > 1) t += 1.0;                                                       // in-LDS sum, half of the waves idle in the epilogue
    return t;
}

static int best_split(long blocks, int bm, int bn, int nk, bool tinyM, int occ, int kg, int M, int N, double* t_out) {
    const int smax = tinyM ? 128 : 32, min_chunks = (tinyM ? 2 : 4) * kg;
    int best = 1; double bt = 1e30;
    for (int s = 1; s <= smax && s <= (nk >= min_chunks ? nk / min_chunks : 1); ++s) {
        double t;
        if (tinyM) { const long g = blocks * s; t = g >= 512 ? 1.0 + 1e-4 * s : 512.0 / (double)g; }   // weight streaming: two workgroups per CU keep enough loads in flight (1024 measured 30 % slower: more slabs, same bandwidth)
        else t = est_time(blocks, bm, bn, s, occ, kg, M, N, nk);
        if (t < bt - 1e-9) { bt = t; best = s; }
    }
    if (t_out) *t_out = bt;
    return best;
}


static void choose_igemm(const Layer& l, int M, int nphases, int nk, int* tile, int* splits, int* kg, int bf16 = 0 /* 1 bf16, 2 split-bf16 */) {
    const bool kg_auto = *kg == 0, splits_auto = *splits == 0;
    // bf16: 64 channels per chunk; the kernel is bound by the operand movement, not the matrix pipe: price
    // the main loop at the measured rate so that the fixed costs (launch, stores, slabs) weigh as they do in the measurements
    g_plan_peak = bf16 == 2 ? g_x3_peak : bf16 ? 0.86e9 : 157.3e6; g_plan_ck = bf16 ? 64 : 32;   // 128x128 bf16 tiles measured ~620 TF = 0.72 x 860
    g_plan_kg2 = bf16 == 2 ? g_x3_kg2 : 1.0;
    // MFMA-bound layers: every CU should hold a full set of co-resident wave groups for the whole launch.  Candidate
    // M-tiles 128 / 96 / 64 (N-tile 128), with one or two K groups per workgroup and the best external split-K, are
    // compared by estimated time.
    const bool kg_ok = M > 64 && nk >= 16;
    const int N = l.cout;
    if (*tile == 0) {
        if (N % 128 != 0) *tile = TILE_128x64;                         // (128x64 measured 94 TF vs 79 TF for the old 256x64 tile)
        else if (M <= 32) *tile = TILE_32x128;
        else if (M <= 64) *tile = TILE_64x128;
        else {
            const int cand[3] = {TILE_128x128, TILE_96x128, TILE_64x128};
            const double bias[3] = {1.0, 1.02, 1.08};                     // smaller tiles re-read more B per flop
            double bt = 1e30; int btile = TILE_128x128, bk = 1;
            for (int c = 0; c < 3; ++c)
                for (int k = 1; k <= ((kg_ok && *kg == 0) ? 2 : 1); ++k) {
                    const int kk = *kg > 0 ? *kg : k;
                    int bm, bn; tile_dims(cand[c], &bm, &bn);
                    const long mt = (M + bm - 1) / bm;
                    double t = 0.0;
                    best_split(mt * (N / bn) * nphases, bm, bn, nk, false, tile_occ(cand[c], kk), kk, M, N, &t);
                    // fp32: smaller tiles re-read more B per flop; bf16: the kernel is bound by the operand movement, time
                    // scales with operand bytes per flop, (1/BM + 1/BN)
                    t *= bf16 ? (1.0 / bm + 1.0 / bn) * 64.0 : bias[c];
                    if (t < bt - 1e-9) { bt = t; btile = cand[c]; bk = kk; }
                }
            *tile = btile;
            if (*kg == 0) *kg = bk;
        }
    }
    int bm, bn; tile_dims(*tile, &bm, &bn);
    const long blocks = (long)((M + bm - 1) / bm) * (N / bn) * nphases;
    if (*kg == 0) {
        *kg = 1;
        if (kg_ok && bm <= 128 && *splits == 0) {
            double t1 = 0.0, t2 = 0.0;
            best_split(blocks, bm, bn, nk, false, tile_occ(*tile, 1), 1, M, N, &t1);
            best_split(blocks, bm, bn, nk, false, tile_occ(*tile, 2), 2, M, N, &t2);
            if (t2 < t1 - 1e-9) *kg = 2;
        }
    }
    if (*splits == 0) *splits = best_split(blocks, bm, bn, nk, M <= 64, tile_occ(*tile, *kg), *kg, M, N, nullptr);
    // Two K groups in one workgroup (8 waves, 125 KiB of LDS: nothing else fits beside it on a CU) exist to save slabs and reduce work.  Where the plan
    // needs an external split anyway and the external-split-only form (four-wave workgroups of 62 KiB: two of ANY two launches share a CU, which is what
    // the window on the other lane needs) is estimated within one per cent, take that form: a tie alone, and measured under two lanes (round 5,
    // profiles/r05/r_plan_ab_n300.txt) decoder c3 -- the one layer this selects at 300 frames -- moves the step 1.1015 -> 1.0767 ms per window and ends
    // the bimodal phase lock of the lanes; encoder c4 / c5 and decoder c2 (4 - 12 % apart by the estimate) gain nothing and stay.
    if (kg_auto && splits_auto && bf16 == 0 && *kg == 2 && *splits > 1 && M > 64) {
        double t1 = 0.0;
        const int s1 = best_split(blocks, bm, bn, nk, false, tile_occ(*tile, 1), 1, M, N, &t1);
        const double t2 = est_time(blocks, bm, bn, *splits, tile_occ(*tile, 2), 2, M, N, nk);
        if (t1 <= 1.01 * t2) { *kg = 1; *splits = s1; }
    }
    if (*splits * *kg > nk) { *kg = 1; if (*splits > nk) *splits = nk; }
}

// ---- stage-2 output-stationary layers (ry_c2d_os) ----
// (MT4, NT4, WAVES, DEPTH): tile of 4 MT4 rows x 4 NT4 output channels per workgroup, WAVES waves that deal the K units among them in rounds
// of four, DEPTH units in flight per wave.  Sixteen-wave workgroups have 128 registers per lane: small tiles only.
#define RY_OS2_CONFIGS(X)                                                                                              \
    X(1, 1, 4, 4) X(1, 1, 8, 4) X(1, 1, 8, 2) X(1, 1, 16, 2) X(2, 1, 4, 4) X(2, 1, 8, 4) X(2, 1, 8, 2) X(2, 1, 16, 2)     \
    X(3, 1, 4, 4) X(3, 1, 8, 4) X(3, 1, 8, 2) X(3, 1, 16, 2) X(4, 1, 4, 4) X(4, 1, 8, 4) X(4, 1, 8, 2) X(4, 1, 16, 2)     \
    X(6, 1, 4, 4) X(6, 1, 8, 2)                                                             \
    X(1, 2, 4, 4) X(1, 2, 8, 4) X(1, 2, 8, 2) X(1, 2, 16, 2) X(2, 2, 4, 4) X(2, 2, 8, 4) X(2, 2, 8, 2) X(2, 2, 16, 2)     \
    X(3, 2, 4, 4) X(3, 2, 8, 4) X(3, 2, 8, 2) X(3, 2, 16, 2) X(4, 2, 4, 4) X(4, 2, 8, 4) X(4, 2, 8, 2) X(4, 2, 16, 2)     \
    X(6, 2, 4, 4) X(6, 2, 8, 2)                                                                             \
    X(1, 4, 4, 4) X(1, 4, 8, 4) X(1, 4, 8, 2) X(1, 4, 16, 2) X(2, 4, 4, 4) X(2, 4, 8, 4) X(2, 4, 8, 2) X(2, 4, 16, 2)     \
    X(3, 4, 4, 4) X(3, 4, 8, 4) X(3, 4, 8, 2) X(4, 4, 4, 4) X(4, 4, 8, 2) X(6, 4, 4, 2)

static bool os2_has_config(int mt4, int nt4, int waves, int depth) {
#define X(A, B, C, D) if (mt4 == A && nt4 == B && waves == C && depth == D) return true;
    RY_OS2_CONFIGS(X)
#undef X
    return false;
}

// The LDS-DMA pixel path keeps one KiB per (wave, four tile rows, unit in flight): slices with two units in flight and at most 64 KiB of ring
// (what the other window lane's kernels leave free on a CU).
static constexpr bool os2_xl_ok(int mt4, int waves, int depth) { return depth == 2 && mt4 * waves <= 32; }
static int g_os2_maxcost = 4608;       // RY_OS2_MAXCOST: a layer with the ry_c2d_os filter layout runs output-stationary when slice cost x K units stays below this (0: never).
                                       // Fitted: encoder c6 / decoder c1 at 300 frames (4096) win by 2-4 us, encoder c5 at 300 frames (10240) and decoder c2 at 100 frames (8192) lose by 8-11
static int g_os2_force[16][4];         // RY_OS2="layer:mt4:nt4:waves:depth,...": tuning aid, fixes the slice of single layers ("layer:0" keeps that layer on the implicit GEMM)
static bool g_os2_forced[16];

// Slice of one layer, by a cost fitted to the slice sweeps on the MI355X (profiles/r05/e_os_sweep_n{300,100}.txt): a workgroup pulls K x (rows +
// channels) of its tile through its CU's L1, the pixel rows at about half the rate of the filter rows (a wave-load of pixels is four
// 256-byte pieces of four different pixels, a wave-load of filters one contiguous KiB), and the launch takes as many rounds as there are
// workgroups per CU.  cost = max(1, workgroups / 256) x (2 rows + channels) of the tile (1.5 rows where the pixels travel by DMA); the sweeps rank the slices of every bottom layer in this
// order (encoder c7: 4 x 8 < 8 x 4 < 4 x 4 < 12 x 4; decoder c1: 24 x 16 < 12 x 16 < 8 x 16 < 16 x 16).  `cost_out` x K units is what the
// caller compares with the implicit GEMM (g_os2_maxcost).
static bool choose_os2(int M, int N, int nphases, int U, int* mt4, int* nt4, int* waves, int* depth, double* cost_out = nullptr) {
    static const int MTS[5] = {1, 2, 3, 4, 6}, NTS[3] = {1, 2, 4};
    double best = 1e30; int bm = 0, bn = 0, bw = 0, bd = 0;
    for (int mi = 0; mi < 5; ++mi)
        for (int ni = 0; ni < 3; ++ni) {
            const int m = MTS[mi], n = NTS[ni];
            if ((*mt4 != 0 && *mt4 != m) || (*nt4 != 0 && *nt4 != n) || N % (4 * n) != 0) continue;
            // waves x units in flight: large tiles run four waves with four units in flight, small ones eight waves with two (more waves hide more of
            // the chain request -> landing -> MFMA at one or two workgroups per CU); whatever the run of K units feeds with whole rounds of four
            int w = 0, d = 0;
            static const int WD[4][2] = {{8, 2}, {4, 4}, {16, 2}, {8, 4}}, WD_BIG[4][2] = {{4, 4}, {8, 2}, {4, 2}, {8, 4}};
            const bool big = m * n >= 12 && !os2_xl_ok(m, 8, 2);        // (a large tile whose pixels go through the LDS keeps eight waves)
            for (int k = 0; k < 4 && w == 0; ++k) {
                const int cw = (big ? WD_BIG : WD)[k][0], cd = (big ? WD_BIG : WD)[k][1];
                if ((*waves != 0 && *waves != cw) || (*depth != 0 && *depth != cd) || U % (4 * cw) != 0 || !os2_has_config(m, n, cw, cd)) continue;
                w = cw; d = cd;
            }
            if (w == 0 && *waves != 0 && *depth != 0)                    // a forced pair outside the preference lists (only one of the two forced and no preferred pair fits: no plan)
                if (U % (4 * *waves) == 0 && os2_has_config(m, n, *waves, *depth)) { w = *waves; d = *depth; }
            if (w == 0) continue;
            const double wgs = (double)((M + 4 * m - 1) / (4 * m)) * (N / (4 * n)) * nphases;
            const double px = os2_xl_ok(m, w, d) ? 6.0 : 8.0;      // pixel rows by DMA through the LDS: contiguous 256-byte pieces (encoder c6: 12 x 8 ahead of 8 x 16)
            const double cost = (wgs > 256.0 ? wgs / 256.0 : 1.0) * (px * m + 4.0 * n) * ((double)((M + 4 * m - 1) / (4 * m)) * 4 * m / M);   // padded rows are loaded too
            if (cost < best - 1e-9) { best = cost; bm = m; bn = n; bw = w; bd = d; }
        }
    if (bm == 0) return false;
    *mt4 = bm; *nt4 = bn; *waves = bw; *depth = bd;
    if (cost_out) *cost_out = best;
    return true;
}

static int launch_c2d_os(Launcher& Lc, const Layer& l, const LayerPlan& lp, int B, const float* s1, int C1, const float* s2, int C2, float slope) {
    const TapTable t = make_taps(l);
    RyC2dOsParams p;
    memset(&p, 0, sizeof p);
    p.src1 = s1; p.src2 = s2; p.wt = l.w2os; p.scale = l.scale; p.shift = l.shift;
    p.out = lp.w32 ? lp.out : nullptr; p.out16 = lp.w16 ? lp.out16 : nullptr; p.x3 = lp.o16x3 ? 1 : 0;
    p.C1 = C1; p.C2 = C2; p.B = B; p.Hi = lp.Hi; p.Wi = lp.Wi; p.Ho = lp.Ho; p.Wo = lp.Wo;
    if (l.deconv) { p.Mh = lp.Hi; p.Mw = lp.Wi; p.stride = 1; p.pad = 0; p.ostride = 2; }
    else { p.Mh = lp.Ho; p.Mw = lp.Wo; p.stride = l.stride; p.pad = l.pad; p.ostride = 1; }
    p.M = B * p.Mh * p.Mw;
    p.ntaps = t.ntaps; p.nphases = t.nphases; p.N = l.cout; p.act = l.act; p.slope = slope;
    const int MT = 4 * lp.os2_mt4, NT = 4 * lp.os2_nt4;
    p.mtiles = (p.M + MT - 1) / MT; p.ntiles = l.cout / NT;
    const int cpt = (C1 + C2) / 64, U = t.ntaps * cpt;
    if (!l.w2os || C1 % 256 || C2 % 256 || l.cout % NT || U % (4 * lp.os2_waves) || (size_t)C1 > ZTAIL || (size_t)C2 > ZTAIL)
        return fail(RY_ESTATE, "%s: not a shape for the output-stationary kernel (slice %dx%d, %d waves, depth %d)", l.name, lp.os2_mt4, lp.os2_nt4, lp.os2_waves, lp.os2_depth);
    if (p.M >= (1 << 24) || (long long)p.mtiles * p.ntiles * p.nphases >= (1 << 24)) return fail(RY_EINVAL, "%s: more than 2^24 rows or tiles in one launch", l.name);
    if (((size_t)B * lp.Hi * lp.Wi * (size_t)(C1 > C2 ? C1 : C2) + ZTAIL) * 4 >= ((size_t)1 << 32))      // the kernel's pixel offsets (zp1 / zp2, its offset table) are 32-bit byte offsets
        return fail(RY_EINVAL, "%s: a source of 4 GiB or more does not fit the output-stationary kernel's 32-bit offsets", l.name);
    p.zp1 = (unsigned)((size_t)B * lp.Hi * lp.Wi * C1 * 4); p.zp2 = (unsigned)((size_t)B * lp.Hi * lp.Wi * C2 * 4);
    p.inv_Mimg = 1.f / (float)(p.Mh * p.Mw); p.inv_Mw = 1.f / p.Mw; p.inv_mtiles = 1.f / p.mtiles; p.inv_ntiles = 1.f / p.ntiles; p.inv_cpt = 1.f / cpt;
    p.kw = l.deconv ? 2 : l.k; p.dil = l.deconv ? 1 : l.dil; p.inv_kw = 1.f / p.kw;
    const int total = p.mtiles * p.ntiles * p.nphases;
    dim3 grid((unsigned)(((total + 7) / 8) * 8));
    char nm[48];
    const bool xl = os2_xl_ok(lp.os2_mt4, lp.os2_waves, lp.os2_depth);
    snprintf(nm, sizeof nm, "ry_c2d_os<%d,%d,%d,%d,%s>", lp.os2_mt4, lp.os2_nt4, lp.os2_waves, lp.os2_depth, xl ? "true" : "false");      // as rocprofv3 prints it
    RY_TRY(Lc.begin(nm, l.name, lp.flops, lp.bytes, grid));
    bool done = false;
#define X(A, B_, C, D)                                                                                          \
    if (!done && lp.os2_mt4 == A && lp.os2_nt4 == B_ && lp.os2_waves == C && lp.os2_depth == D) {               \
        RY_LAUNCH((ry_c2d_os<A, B_, C, D, os2_xl_ok(A, C, D)>), grid, 64 * C, Lc.stream, p);                    \
        done = true;                                                                                            \
    }
    RY_OS2_CONFIGS(X)
#undef X
    if (!done) return fail(RY_EINVAL, "%s: no ry_c2d_os instantiation <%d,%d,%d,%d>", l.name, lp.os2_mt4, lp.os2_nt4, lp.os2_waves, lp.os2_depth);
    return Lc.end();
}

// the rows a launch left out of its grid (LayerPlan::hole_*): copies of the row above them, into the fp32 output and / or the bf16 copy ([pixel][N] or split [pixel][hi | lo])
static int launch_rep_rows(Launcher& Lc, const Layer& l, const LayerPlan& lp, int B) {
    for (int copy = 0; copy < 2; ++copy) {
        if (copy == 0 ? !lp.w32 : !lp.w16) continue;
        const int px_bytes = copy == 0 ? 4 * l.cout : (lp.o16x3 ? 4 : 2) * l.cout;
        RyRepRowsParams q;
        q.base = copy == 0 ? lp.out : reinterpret_cast<float*>(lp.out16);
        q.row_f4 = lp.Wo * px_bytes / 16; q.img_f4 = (long long)lp.Ho * q.row_f4;
        q.src = lp.hole_lo - 1; q.dst0 = lp.hole_lo; q.nrows = lp.hole_n;
        dim3 rg((unsigned)((q.row_f4 + 255) / 256), (unsigned)lp.hole_n, (unsigned)B);
        RY_TRY(Lc.begin("ry_rep_rows", l.name, 0, (double)B * lp.hole_n * lp.Wo * px_bytes, rg));
        RY_LAUNCH(ry_rep_rows, rg, 256, Lc.stream, q);
        RY_TRY(Lc.end());
    }
    return RY_OK;
}

// sum of the raw split-K slabs of a launch + folded BN + activation (Ho_run = the output rows per image the launch covered, oo = their float offset)
static int launch_reduce(Launcher& Lc, const Layer& l, const LayerPlan& lp, int B, int Ho_run, size_t oo, long long slab_stride, float slope) {
    RyReduceParams r;
    const size_t ro = B == 1 ? oo : 0;                                    // one window: only the rows this launch wrote
    r.slabs = lp.slabs + ro; r.splits = lp.splits; r.slab_stride = slab_stride;
    r.scale = l.scale; r.shift = l.shift; r.out = lp.w32 ? lp.out + ro : nullptr; r.out16 = lp.w16 ? lp.out16 + ro * (lp.o16x3 ? 2 : 1) : nullptr;
    r.x3 = lp.o16x3 ? 1 : 0;
    r.total = B == 1 ? (long long)Ho_run * lp.Wo * l.cout : slab_stride; r.N = l.cout;      // one window: only the rows this launch wrote (a prefix when cropped)
    r.act = l.act; r.slope = slope;
    if (lp.splits >= 16 && r.total <= (1 << 20)) {      // many slabs, few outputs
        dim3 rg((unsigned)((r.total / 4 + 63) / 64));
        RY_TRY(Lc.begin("ry_splitk_reduce_wide", l.name, 0, (double)r.total * 4 * (lp.splits + 1), rg));
        RY_LAUNCH(ry_splitk_reduce_wide, rg, 256, Lc.stream, r);
    } else {
        dim3 rg((unsigned)((r.total / 4 + 255) / 256));
        RY_TRY(Lc.begin("ry_splitk_reduce", l.name, 0, (double)r.total * 4 * (lp.splits + 1), rg));
        RY_LAUNCH(ry_splitk_reduce, rg, 256, Lc.stream, r);
    }
    return Lc.end();
}

// ---- stage-2 layers in Winograd F(2x2, 2x2) form (ry_wino_ldsdma) ----
// Workgroup shapes: cfg 1 = 2 x 2 waves (two M-blocks of 8 x 16 pixels x 64 channels, one 8-channel slice per iteration, 74 KiB of LDS: two per CU),
// cfg 2 = 4 x 2 waves (four M-blocks x 64 channels, two slices per iteration, 146 KiB: one per CU).  mbw = M-blocks per tile row.
static int g_wino = 1;                 // RY_WINOGRAD=0: every layer keeps the direct implicit GEMM (the bit-exact reference of the Winograd form; A/B)
static int g_wino_min_m = 512;         // RY_WINO_MINM: rows (pixels of one phase) from which an eligible layer takes the Winograd form
static int g_wino_force[16][3];        // RY_WINO="layer:cfg:mbw:splits,...": tuning aid, fixes the Winograd plan of single layers ("layer:0" keeps that layer on the direct kernel)
static bool g_wino_forced[16];

static bool wino_cfg_dims(int cfg, int* wm, int* wn, int* nsl) {
    if (cfg == 1) { *wm = 2; *wn = 2; *nsl = 1; return true; }
    if (cfg == 2) { *wm = 4; *wn = 2; *nsl = 2; return true; }
    return false;
}
static void wino_tile_hw(int cfg, int mbw, int* th, int* tw) {              // pixels of the stencil's output grid per M-tile
    int wm = 2, wn = 2, nsl = 1; wino_cfg_dims(cfg, &wm, &wn, &nsl);
    *th = 8 * (wm / mbw); *tw = 16 * mbw;
}
static const char* wino_name(int cfg, int mode) {
    static char buf[4][40];
    char* b = buf[(cfg - 1) * 2 + (mode - 1)];
    int wm = 2, wn = 2, nsl = 1; wino_cfg_dims(cfg, &wm, &wn, &nsl);
    snprintf(b, 40, "ry_wino_ldsdma<%d,%d,%d,%d>", wm, wn, nsl, mode);      // as rocprofv3 prints it
    return b;
}

// Plan of one layer: workgroup shape, tile shape (the squarest one that divides the grid: the patch carries one extra row and column), external split-K.
// Returns false when no tile shape divides the Mh x Mw grid.
static bool choose_wino(int Mh, int Mw, int N, int nphases, int npatches, int B, int* cfg, int* mbw, int* splits) {
    double best = 1e30; int bc = 0, bm = 0, bs = 0;
    for (int c = 1; c <= 2; ++c) {
        if (*cfg != 0 && *cfg != c) continue;
        int wm, wn, nsl; wino_cfg_dims(c, &wm, &wn, &nsl);
        for (int m = 1; m <= wm; m *= 2) {
            if (*mbw != 0 && *mbw != m) continue;
            int th, tw; wino_tile_hw(c, m, &th, &tw);
            if (Mh % th || Mw % tw) continue;
            const long units = (long)B * (Mh / th) * (Mw / tw) * (N / 64) * nphases;
            const int slots = c == 1 ? 512 : 256;
            const double halo = (double)(th + 1) * (tw + 1) / ((double)th * tw) + 0.002 * th;      // (short tiles: the dead-row crop and the copied padding rows round to whole tile rows)
            for (int sp = 1; sp <= 32 && sp <= npatches; ++sp) {
                if (*splits != 0 ? *splits != sp : (sp > 1 && sp > npatches / 2)) continue;      // (the planner's own splits leave two patches per workgroup)
                // rounds of workgroups x iterations of the longest split (+ prologue / epilogue of a workgroup, in iterations) x time of an iteration relative to
                // cfg 1 (cfg 2 runs twice the slices on twice the rows per slot), + the slab traffic and the reduce node of an external split
                const long rounds = (units * sp + slots - 1) / slots;
                const double its = (double)((npatches + sp - 1) / sp) * 2.0;          // 8-channel slices
                double t = (double)rounds * (its + 6.0) * (c == 1 ? 1.0 : 2.0) * (0.9 + 0.1 * halo);
                if (sp > 1) t += 8.0 + 0.02 * sp * (double)B * Mh * Mw * nphases * N / 65536.0;
                if (t < best - 1e-9) { best = t; bc = c; bm = m; bs = sp; }
            }
        }
    }
    if (bc == 0) return false;
    *cfg = bc; *mbw = bm; *splits = bs;
    return true;
}

static int launch_wino(Launcher& Lc, const Layer& l, const LayerPlan& lp, const float* wwin, int B, const float* s1, int C1, const float* s2, int C2, float slope) {
    RyConvGeom g;
    fill_geom(g, l, lp, B, s1, C1, s2, C2);
    int wm, wn, nsl;
    if (!wwin || !wino_eligible(l, 2) || !wino_cfg_dims(lp.wino_cfg, &wm, &wn, &nsl) || lp.wino_mbw < 1 || lp.wino_mbw > wm || (wm % lp.wino_mbw))
        return fail(RY_ESTATE, "%s: not a layer / plan for the Winograd kernel (cfg %d, %d blocks per tile row)", l.name, lp.wino_cfg, lp.wino_mbw);
    int th, tw; wino_tile_hw(lp.wino_cfg, lp.wino_mbw, &th, &tw);
    if (g.Mh % th || g.Mw % tw) return fail(RY_ESTATE, "%s: the %d x %d grid is not a multiple of the %d x %d Winograd tile", l.name, g.Mh, g.Mw, th, tw);
    RyWinoParams p;
    memset(&p, 0, sizeof p);
    p.g = g; p.wt = wwin; p.scale = l.scale; p.shift = l.shift;
    p.splits = lp.splits; p.act = l.act; p.slope = slope;
    p.slab_stride = (long long)B * lp.Ho * lp.Wo * l.cout;
    const size_t oo = lp.crop_hi > 0 ? (size_t)(l.deconv ? 2 : 1) * lp.crop_lo * lp.Wo * l.cout : 0;     // output rows start (2 x for the sub-pixel form) crop_lo rows into every image
    p.out = lp.splits > 1 ? lp.slabs + oo : lp.out + oo;
    if (!p.out) return fail(RY_ESTATE, "%s: no output buffer", l.name);
    p.mbw = lp.wino_mbw; p.tcols = g.Mw / tw; p.trows = g.Mh / th;
    p.hole_ty = 1 << 30; p.hole_nt = 0;
    if (lp.hole_n > 0) {
        if (l.deconv || lp.crop_hi > 0 || lp.hole_lo % th || lp.hole_n % th || lp.hole_lo + lp.hole_n > g.Mh)
            return fail(RY_ESTATE, "%s: rows %d..%d cannot be left out of this launch", l.name, lp.hole_lo, lp.hole_lo + lp.hole_n - 1);
        p.hole_ty = lp.hole_lo / th; p.hole_nt = lp.hole_n / th; p.trows -= p.hole_nt;
    }
    p.mtiles = B * p.trows * p.tcols; p.ntiles = l.cout / (32 * wn);
    p.npatches = (l.deconv ? 1 : 4) * ((C1 + C2) / 16);
    if (lp.splits < 1 || lp.splits > p.npatches) return fail(RY_ESTATE, "%s: %d splits for %d patches", l.name, lp.splits, p.npatches);
    p.kq = p.npatches / lp.splits; p.krem = p.npatches % lp.splits;
    const int nsl_ = lp.splits * p.ntiles * g.nphases;
    if ((long long)p.mtiles * nsl_ >= (1 << 24) || (long long)B * g.Mh * g.Mw >= (1 << 24)) return fail(RY_EINVAL, "%s: more than 2^24 output rows or tiles in one launch; lower the batch", l.name);
    p.inv_nphases = 1.f / g.nphases; p.inv_ntiles = 1.f / p.ntiles; p.inv_tcols = 1.f / p.tcols; p.inv_trows = 1.f / p.trows; p.inv_pw = 1.f / (float)(tw + 1);
    p.inv_nsl = 1.f / nsl_;
    p.xcd_gs = 0; p.xcd_gs_shift = 0; p.xcd_nsg = 1; p.xcd_mtg = 1; p.inv_xcd_nsg = 1.f;
    {   // XCD grouping as the implicit GEMM: gm M-tile groups x gs slice groups, the split with the least L2 miss traffic among those that divide evenly
        const double wbytes = 2.25 * g.nphases * l.cout * 4.0 * (C1 + C2), abytes = (double)B * lp.Hi * lp.Wi * (C1 + C2);
        double best = 1e300;
        for (int sh = 0; sh <= 3; ++sh) {
            const int gs = 1 << sh, gm = 8 >> sh;
            if (nsl_ % gs != 0 || p.mtiles % gm != 0) continue;
            const double cost = gm * wbytes + gs * abytes;
            if (cost < best) { best = cost; p.xcd_gs = gs; p.xcd_gs_shift = sh; p.xcd_nsg = nsl_ / gs; p.xcd_mtg = p.mtiles / gm; }
        }
        if (p.xcd_gs) p.inv_xcd_nsg = 1.f / p.xcd_nsg;
    }
    const int total_tiles = p.mtiles * nsl_;
    dim3 grid((unsigned)(((total_tiles + 7) / 8) * 8));
    const int mode = l.deconv ? 1 : 2;
    RY_TRY(Lc.begin(wino_name(lp.wino_cfg, mode), l.name, lp.flops, lp.bytes, grid, lp.flops * 9.0 / 16.0));
    if (lp.wino_cfg == 1) {
        if (mode == 1) RY_LAUNCH((ry_wino_ldsdma<2, 2, 1, 1>), grid, 256, Lc.stream, p);
        else RY_LAUNCH((ry_wino_ldsdma<2, 2, 1, 2>), grid, 256, Lc.stream, p);
    } else {
        if (mode == 1) RY_LAUNCH((ry_wino_ldsdma<4, 2, 2, 1>), grid, 512, Lc.stream, p);
        else RY_LAUNCH((ry_wino_ldsdma<4, 2, 2, 2>), grid, 512, Lc.stream, p);
    }
    RY_TRY(Lc.end());
    // (with an external split the rows left out of the grid have no slabs: the reduce node writes whatever their slab memory holds, the copy node behind it fills them in)
    if (lp.splits > 1) RY_TRY(launch_reduce(Lc, l, lp, B, g.Ho, oo, p.slab_stride, slope));
    if (p.hole_nt > 0) RY_TRY(launch_rep_rows(Lc, l, lp, B));
    return RY_OK;
}

static int launch_conv2d(Launcher& Lc, const Layer& l, const LayerPlan& lp, int B, const float* s1, int C1, const float* s2, int C2, float slope) {
    if (lp.path == PATH_OS2D) return launch_c2d_os(Lc, l, lp, B, s1, C1, s2, C2, slope);
    if (lp.path == PATH_WINO) return launch_wino(Lc, l, lp, l.wwin, B, s1, C1, s2, C2, slope);
    RyConvGeom g;
    fill_geom(g, l, lp, B, s1, C1, s2, C2);
    const int M = B * g.Mh * g.Mw;
    if (lp.path == PATH_IGEMM || lp.path == PATH_IGEMM_BF16) {
        const bool bf16 = lp.path == PATH_IGEMM_BF16;       // s1 / s2 then point to bf16 activations
        if (bf16 && lp.x3) { C1 *= 3; C2 *= 3; }            // split-bf16: the K axis the kernel walks (g.C1 / g.C2)
        RyIgemmParams p;
        p.g = g; p.wt = bf16 ? (lp.x3 ? l.wigx3 : l.wig16) : l.wig; p.scale = l.scale; p.shift = l.shift;
        p.x3 = lp.o16x3 ? 1 : 0;
        p.splits = lp.splits; p.act = l.act; p.slope = slope;
        p.slab_stride = (long long)B * lp.Ho * lp.Wo * l.cout;
        // output rows start (2 x for the sub-pixel form) crop_lo rows into every image
        const size_t oo = lp.crop_hi > 0 ? (size_t)(l.deconv ? 2 : 1) * lp.crop_lo * lp.Wo * l.cout : 0;
        p.out = lp.splits > 1 ? lp.slabs + oo : (lp.w32 ? lp.out + oo : nullptr);
        p.out16 = (lp.splits == 1 && lp.w16) ? lp.out16 + oo * (lp.o16x3 ? 2 : 1) : nullptr;
        int bm, bn; tile_dims(lp.tile, &bm, &bn);
        p.mtiles = (M + bm - 1) / bm; p.ntiles = l.cout / bn;
        p.tw = 0;
        for (int tw = 16; tw >= 4; tw >>= 1)           // 2-D M-tiles when the row grid divides evenly, else BM consecutive rows in raster order
            if (bm % tw == 0 && g.Mw % tw == 0 && g.Mh % (bm / tw) == 0) { p.tw = tw; break; }
        p.hole_ty = 1 << 30; p.hole_nt = 0;
        if (lp.hole_n > 0) {                            // whole tile rows inside the stretch of identical padding rows are left out (ry_rep_rows fills them in)
            const int th = p.tw > 0 ? bm / p.tw : 0;
            if (p.tw == 0 || l.deconv || lp.splits != 1 || lp.crop_hi > 0 || lp.hole_lo % th || lp.hole_n % th || lp.hole_lo + lp.hole_n > g.Mh)
                return fail(RY_ESTATE, "%s: rows %d..%d cannot be left out of this launch", l.name, lp.hole_lo, lp.hole_lo + lp.hole_n - 1);
            p.hole_ty = lp.hole_lo / th; p.hole_nt = lp.hole_n / th;
            p.mtiles -= B * p.hole_nt * (g.Mw / p.tw);
        }
        int patch = 0;
        {   // prologue helpers of the LDS-DMA kernel (ry_fdiv reciprocals; operands stay below 2^24, checked here)
            const int ck = bf16 ? 64 : 32, cpt = (C1 + C2) / ck, nkc = g.ntaps * cpt;
            if ((long long)p.mtiles * p.ntiles * g.nphases * lp.splits >= (1 << 24) || M >= (1 << 24))
                return fail(RY_EINVAL, "%s: more than 2^24 output rows or tiles in one launch; lower the batch", l.name);
            p.inv_nphases = 1.f / g.nphases; p.inv_ntiles = 1.f / p.ntiles; p.inv_mtiles = 1.f / p.mtiles;
            // XCD grouping: gm M-tile groups x gs slice groups (gm * gs = 8 L2s); every filter byte is fetched by gm L2s, every
            // input byte by gs -- pick the split with the least L2 miss traffic among those that divide evenly
            const int nsl = lp.splits * p.ntiles * g.nphases;
            p.inv_nsl = 1.f / nsl;
            p.xcd_gs = 0; p.xcd_gs_shift = 0; p.xcd_nsg = 1; p.xcd_mtg = 1; p.inv_xcd_nsg = 1.f;
            {
                const double wbytes = (double)g.nphases * l.cout * g.ntaps * (C1 + C2), abytes = (double)B * lp.Hi * lp.Wi * (C1 + C2);
                double best = 1e300;
                for (int sh = 0; sh <= 3; ++sh) {
                    const int gs = 1 << sh, gm = 8 >> sh;
                    if (nsl % gs != 0 || p.mtiles % gm != 0) continue;
                    const double cost = gm * wbytes + gs * abytes;
                    if (cost < best) { best = cost; p.xcd_gs = gs; p.xcd_gs_shift = sh; p.xcd_nsg = nsl / gs; p.xcd_mtg = p.mtiles / gm; }
                }
                if (p.xcd_gs) p.inv_xcd_nsg = 1.f / p.xcd_nsg;
            }
            p.inv_Mimg = 1.f / (float)(g.Mh * g.Mw); p.inv_Mw = 1.f / g.Mw; p.inv_cpt = 1.f / cpt; p.inv_kw = 1.f / g.kw;
            p.tw_shift = 0; p.th = 1; p.tcols = 1; p.trows = 1; p.inv_tcols = 1.f; p.inv_trows = 1.f;
            if (p.tw > 0) {
                while ((1 << p.tw_shift) < p.tw) ++p.tw_shift;
                p.th = bm / p.tw; p.tcols = g.Mw / p.tw; p.trows = g.Mh / p.th - p.hole_nt;
                p.inv_tcols = 1.f / p.tcols; p.inv_trows = 1.f / p.trows;
            }
            // sub-pixel deconvolution on 16-pixel-wide 2-D tiles: the patch variant of the kernel (K units = whole channel chunks)
            // 1: sub-pixel deconvolution, one patch per channel chunk; 2: k4 s2 p1 convolution, one patch per (chunk, input parity)
            if (p.tw == 16 && (M >= 512 || lp.any_m_patch)) {   // small layers: the longer set-up costs more than the reuse saves (measured at M = 192)
                if (g.ostride == 2 && (g_patch & 1) && lp.splits * lp.kg <= cpt) patch = 1;
                else if (!l.deconv && l.k == 4 && l.stride == 2 && l.pad == 1 && l.dil == 1 && (g_patch & 2) && lp.splits * lp.kg <= 4 * cpt) patch = 2;
            }
            const int units = patch == 1 ? cpt : patch == 2 ? 4 * cpt : nkc;
            p.kq = units / lp.splits; p.krem = units % lp.splits;
        }
        const int total_tiles = p.mtiles * p.ntiles * g.nphases * lp.splits;
        dim3 grid((unsigned)(((total_tiles + 7) / 8) * 8));
        RY_TRY(Lc.begin(tile_name(lp.tile, lp.kg, bf16, patch), l.name, lp.flops, lp.bytes, grid));
    #define RY_IGEMM_LAUNCH(BM_, BN_, WM_, WN_)                                                                    \
    do {                                                                                                    \
        if (patch == 1 && lp.kg == 2) RY_LAUNCH((ry_igemm_ldsdma<BM_, BN_, WM_, WN_, 2, false, 1>), grid, 512, Lc.stream, p); \
        else if (patch == 1) RY_LAUNCH((ry_igemm_ldsdma<BM_, BN_, WM_, WN_, 1, false, 1>), grid, 256, Lc.stream, p);          \
        else if (patch == 2 && lp.kg == 2) RY_LAUNCH((ry_igemm_ldsdma<BM_, BN_, WM_, WN_, 2, false, 2>), grid, 512, Lc.stream, p); \
        else if (patch == 2) RY_LAUNCH((ry_igemm_ldsdma<BM_, BN_, WM_, WN_, 1, false, 2>), grid, 256, Lc.stream, p);          \
        else if (lp.kg == 2) RY_LAUNCH((ry_igemm_ldsdma<BM_, BN_, WM_, WN_, 2, false, 0>), grid, 512, Lc.stream, p);          \
        else RY_LAUNCH((ry_igemm_ldsdma<BM_, BN_, WM_, WN_, 1, false, 0>), grid, 256, Lc.stream, p);                          \
    } while (0)
#define RY_IGEMM16_LAUNCH(BM_, BN_, WM_, WN_)                                                                  \
    do {                                                                                                    \
        if (patch == 1 && lp.kg == 2) RY_LAUNCH((ry_igemm_ldsdma<BM_, BN_, WM_, WN_, 2, true, 1>), grid, 512, Lc.stream, p); \
        else if (patch == 1) RY_LAUNCH((ry_igemm_ldsdma<BM_, BN_, WM_, WN_, 1, true, 1>), grid, 256, Lc.stream, p);          \
        else if (patch == 2 && lp.kg == 2) RY_LAUNCH((ry_igemm_ldsdma<BM_, BN_, WM_, WN_, 2, true, 2>), grid, 512, Lc.stream, p); \
        else if (patch == 2) RY_LAUNCH((ry_igemm_ldsdma<BM_, BN_, WM_, WN_, 1, true, 2>), grid, 256, Lc.stream, p);          \
        else if (lp.kg == 2) RY_LAUNCH((ry_igemm_ldsdma<BM_, BN_, WM_, WN_, 2, true, 0>), grid, 512, Lc.stream, p);          \
        else RY_LAUNCH((ry_igemm_ldsdma<BM_, BN_, WM_, WN_, 1, true, 0>), grid, 256, Lc.stream, p);                          \
    } while (0)
        if (bf16) {
            switch (lp.tile) {
                case TILE_128x128: RY_IGEMM16_LAUNCH(128, 128, 2, 2); break;
                case TILE_96x128: RY_IGEMM16_LAUNCH(96, 128, 1, 4); break;
                case TILE_64x128: RY_IGEMM16_LAUNCH(64, 128, 1, 4); break;
                case TILE_128x64: RY_IGEMM16_LAUNCH(128, 64, 4, 1); break;
                default: RY_IGEMM16_LAUNCH(32, 128, 1, 4); break;
            }
#undef RY_IGEMM16_LAUNCH
        } else
        switch (lp.tile) {
            case TILE_128x128: RY_IGEMM_LAUNCH(128, 128, 2, 2); break;
            case TILE_64x128: RY_IGEMM_LAUNCH(64, 128, 1, 4); break;
            case TILE_128x64: RY_IGEMM_LAUNCH(128, 64, 4, 1); break;
            case TILE_96x128: RY_IGEMM_LAUNCH(96, 128, 1, 4); break;
            default: RY_IGEMM_LAUNCH(32, 128, 1, 4); break;
        }
#undef RY_IGEMM_LAUNCH
        RY_TRY(Lc.end());
        if (p.hole_nt > 0) RY_TRY(launch_rep_rows(Lc, l, lp, B));
        if (lp.splits > 1) RY_TRY(launch_reduce(Lc, l, lp, B, g.Ho, oo, p.slab_stride, slope));
    } else if (lp.path == PATH_FIRST) {
        RySrFirstParams p;
        p.x = s1; p.w = l.wdir; p.scale = l.scale; p.shift = l.shift; p.out16 = lp.w16 ? lp.out16 : nullptr;
        p.out = (lp.w32 || !p.out16) ? lp.out : nullptr;          // split-bf16 mode: the fp32 copy only if a consumer reads it
        p.x3 = lp.o16x3 ? 1 : 0;
        p.B = B; p.H = lp.Hi; p.W = lp.Wi; p.N = l.cout; p.act = l.act; p.slope = slope;
        const int quads = l.cout / 4;
        p.qshift = -1;
        for (int sh = 0; sh < 16; ++sh) if ((1 << sh) == quads) p.qshift = sh;
        const long long per_row = (long long)((lp.Wi + 3) / 4) * quads;
        dim3 grid((unsigned)((per_row + 255) / 256), (unsigned)lp.Hi, (unsigned)B);
        if (grid.y > 65535u || grid.z > 65535u) return fail(RY_EINVAL, "%s: %u rows x %u windows exceed the grid limit", l.name, grid.y, grid.z);
        RY_TRY(Lc.begin("ry_sr_first", l.name, lp.flops, lp.bytes, grid));
        RY_LAUNCH((ry_sr_first<4>), grid, 256, Lc.stream, p);
        RY_TRY(Lc.end());
    } else if (lp.path == PATH_LAST) {
        RySrLastParams p;
        p.src1 = s1; p.src2 = s2; p.C1 = C1; p.C2 = C2; p.w = l.wdir; p.scale = l.scale; p.shift = l.shift;
        p.out = lp.out; p.B = B; p.H = lp.Hi; p.W = lp.Wi;
        p.rows_valid = lp.last_rows; p.out_cols = lp.last_cols; p.do_exp = lp.last_exp;
        p.row0 = lp.last_row0; p.out_rows = lp.last_out_rows > 0 ? lp.last_out_rows : lp.last_rows;
        const long long total = (long long)B * p.rows_valid * lp.Wi;
        dim3 grid((unsigned)((total + 7) / 8));
        if (C1 + C2 == 128 && lp.Wi % 16 == 0) {
            const long long strips = (long long)B * p.rows_valid * (lp.Wi / 16);
            p.xcd_band = 1;                                   // every XCD owns a contiguous band of output rows (45.0 -> 31.5 us against raster order)
            const long long nb = (strips + 7) / 8;
            dim3 sg((unsigned)(((nb + 7) / 8) * 8));
            RY_TRY(Lc.begin("ry_sr_last<false>", l.name, lp.flops, lp.bytes, sg));
            p.x3 = 0;
            RY_LAUNCH(ry_sr_last<false>, sg, 256, Lc.stream, p);
        } else {
            RY_TRY(Lc.begin("ry_sr_last_gather", l.name, lp.flops, lp.bytes, grid));
            RY_LAUNCH(ry_sr_last_gather, grid, 256, Lc.stream, p);
        }
        RY_TRY(Lc.end());
    } else {
        RyDirectParams p;
        p.g = g; p.wd = l.wdir; p.scale = l.scale; p.shift = l.shift; p.out = lp.out; p.act = l.act; p.slope = slope;
        const long long total = (long long)M * l.cout;
        dim3 grid((unsigned)((total + 255) / 256), (unsigned)g.nphases);
        RY_TRY(Lc.begin("ry_conv_direct", l.name, lp.flops, lp.bytes, grid));
        RY_LAUNCH(ry_conv_direct, grid, 256, Lc.stream, p);
        RY_TRY(Lc.end());
    }
    return RY_OK;
}

static const int g_s1_wgs = 256, g_s1_maxs = 32;   // split heuristic of the weight-streaming stage-1 kernels (128 / 16 measured 0.275 ms, 256 / 32 0.231 ms, 512 / 64 0.238 ms per forward)

static int c1d_mode(const Layer& l) {
    if (l.deconv) return RY_C1D_DECONV;
    if (l.k == 4 && l.stride == 2 && l.pad == 1 && l.dil == 1) return RY_C1D_S2;
    if (l.stride == 1 && l.dil == 1 && l.pad <= 3) return RY_C1D_S1;
    return RY_C1D_GEN;
}

static int c1d_tile_len(int mode) { return mode == RY_C1D_DECONV ? 8 : 16; }

static int launch_conv1d(Launcher& Lc, const Layer& l, const LayerPlan& lp, int B, const RySrc1d& sa, const RySrc1d& sb, float slope) {
    RyConv1dParams p;
    memset(&p, 0, sizeof p);
    p.s[0] = sa; p.s[1] = sb;
    p.B = B; p.Lin = lp.Wi; p.Lout = lp.Wo; p.Ctot = sa.C + sb.C; p.N = l.cout; p.wd = l.w1d;
    p.stride = l.stride; p.pad = l.pad; p.dil = l.dil;
    p.out = lp.raw; p.splits = lp.splits; p.slab_stride = lp.slab_stride; p.slope = slope;
    const int mode = c1d_mode(l);
    if (mode == RY_C1D_GEN && 15 * l.stride + 3 * l.dil + 1 > 132)
        return fail(RY_EINVAL, "%s: stride %d / dilation %d exceed the staged tile (15*stride + 3*dilation <= 131)", l.name, l.stride, l.dil);
    const int TL = c1d_tile_len(mode);
    const int rows = mode == RY_C1D_DECONV ? lp.Wi : lp.Wo;
    const int tiles = (rows + TL - 1) / TL;
    const int cogroups = (l.cout + 63) / 64;
    const int wpb = cogroups < 4 ? cogroups : 4;                       // waves per workgroup
    dim3 grid((unsigned)((cogroups + 3) / 4), (unsigned)(B * tiles), (unsigned)lp.splits);
    if (grid.y > 65535u) return fail(RY_EINVAL, "%s: batch*tiles = %u exceeds the grid limit", l.name, grid.y);
    const char* nm = mode == RY_C1D_DECONV ? "ry_conv1d_ws<deconv>" : mode == RY_C1D_S2 ? "ry_conv1d_ws<s2>" : mode == RY_C1D_S1 ? "ry_conv1d_ws<s1>" : "ry_conv1d_ws<gen>";
    RY_TRY(Lc.begin(nm, l.name, lp.flops, lp.bytes, grid));
    switch (mode) {
        case RY_C1D_DECONV: RY_LAUNCH((ry_conv1d_ws<RY_C1D_DECONV>), grid, wpb * 64, Lc.stream, p); break;
        case RY_C1D_S2: RY_LAUNCH((ry_conv1d_ws<RY_C1D_S2>), grid, wpb * 64, Lc.stream, p); break;
        case RY_C1D_S1: RY_LAUNCH((ry_conv1d_ws<RY_C1D_S1>), grid, wpb * 64, Lc.stream, p); break;
        default: RY_LAUNCH((ry_conv1d_ws<RY_C1D_GEN>), grid, wpb * 64, Lc.stream, p); break;
    }
    return Lc.end();
}

static int choose_splits_1d(const Layer& l, int B, int rows, int mode) {
    // weight streaming wants many workgroups, but every split is re-summed by each consumer tile: aim for
    // ~256 workgroups, at most 32 splits, at least 16 input channels per split
    const int TL = c1d_tile_len(mode);
    const int cogroups = (l.cout + 63) / 64;
    const long wgs = (long)((cogroups + 3) / 4) * ((rows + TL - 1) / TL) * B;
    int s = (int)((g_s1_wgs + wgs - 1) / wgs);
    const int maxs = l.cin() / 16 > 0 ? l.cin() / 16 : 1;
    if (s > maxs) s = maxs;
    if (s > g_s1_maxs) s = g_s1_maxs;
    if (s < 1) s = 1;
    return s;
}


// ---- stage-1, output-stationary form (ry_c1d_os) ----
static const int g_s1_units = 256;   // smallest workgroup count a layer should reach before it takes a larger slice per workgroup (128 / 256 / 512 / 1024 measured 0.119 / 0.116 / 0.112 / 0.116 ms)

static bool c1d_os_capable(const Layer& l) {
    const int mode = c1d_mode(l);
    return mode != RY_C1D_GEN && l.act != RY_ACT_GLU && l.k <= 4;
}

static int c1d_os_ktw(int ctot) { return ctot <= 64 ? 1 : ctot <= 128 ? 2 : 4; }

// slice (cb output channels x tp rows per position group) of one layer: the largest one that still gives g_s1_units workgroups
static void choose_os(const Layer& l, int B, int rows, int* cb, int* tp) {
    const bool dec = l.deconv;
    static const int CF[4][2] = {{4, 8}, {4, 4}, {2, 8}, {2, 4}};
    const int PG = 4 / c1d_os_ktw(l.cin());
    int best = -1; long best_units = -1; double best_waste = 1e30;
    for (int c = 0; c < 4; ++c) {
        const int b_ = CF[c][0], t_ = CF[c][1];
        if (dec && b_ * t_ * 2 > 32) continue;                                    // 2 tp outputs per input row
        const long tiles = (rows + PG * t_ - 1) / (PG * t_);
        const long units = (long)((l.cout + b_ - 1) / b_) * B * tiles;
        const double waste = (double)tiles * PG * t_ / rows * ((l.cout + b_ - 1) / b_ * b_) / (double)l.cout;
        if (units >= g_s1_units && waste <= 1.34) { best = c; break; }             // candidates are ordered by decreasing slice
        if (units > best_units || (units == best_units && waste < best_waste)) { best = c; best_units = units; best_waste = waste; }
    }
    *cb = CF[best][0]; *tp = CF[best][1];
}

static int launch_c1d_os(Launcher& Lc, const Layer& l, const LayerPlan& lp, int B, const float* sa, int Ca, const float* sb, int Cb,
                         float* out, int keep, float slope, int n_real = 0) {
    RyC1dOsParams p;
    memset(&p, 0, sizeof p);
    p.sa = sa; p.sb = sb; p.Ca = Ca; p.Cb = Cb; p.w = l.w1os; p.scale = l.scale; p.shift = l.shift; p.out = out;
    p.B = B; p.Lin = lp.Wi; p.Lout = lp.Wo; p.N = l.cout; p.keep = keep; p.pad = l.pad; p.act = l.act; p.slope = slope;
    p.kt_shift = lp.os_kt == 4 ? 2 : lp.os_kt == 2 ? 1 : 0; p.n_real = n_real;
    p.dbg = nullptr;                                   // (per-workgroup phase stamps: -DRY_S1_STAMPS diagnostic builds only)
    const int mode = c1d_mode(l);
    const int rows = mode == RY_C1D_DECONV ? lp.Wi : lp.Wo;
    const int PG = 4 / lp.os_kt;
    p.tiles = (rows + PG * lp.os_tp - 1) / (PG * lp.os_tp);
    dim3 grid((unsigned)((l.cout + lp.os_cb - 1) / lp.os_cb), (unsigned)p.tiles, (unsigned)B);
    if (grid.y > 65535u || grid.z > 65535u) return fail(RY_EINVAL, "%s: %u tiles x %u windows exceed the grid limit", l.name, grid.y, grid.z);
    // every wave reads one source when there is no second one or the first ends on a wave boundary (the U-Net's case); else the per-lane form
    const bool usrc = Cb == 0 || Ca % 64 == 0;
    if (!usrc && !(lp.os_cb == 2 && lp.os_tp == 4)) return fail(RY_ESTATE, "%s: a layer whose sources split inside a wave runs the 2x4 slice", l.name);
    char nm[48];
    snprintf(nm, sizeof nm, "ry_c1d_os<%d,%d,%d,%s,%s>", mode, lp.os_cb, lp.os_tp, n_real > 0 ? "true" : "false", usrc ? "true" : "false");   // as rocprofv3 prints it (MODE: 0 = k4 s2 conv, 1 = stride-1 conv, 2 = k4 s2 deconv)
    RY_TRY(Lc.begin(nm, l.name, lp.flops, lp.bytes, grid));
#define RY_OS_CASE(MODE_, CB_, TP_) if (usrc && lp.os_cb == CB_ && lp.os_tp == TP_) { RY_LAUNCH((ry_c1d_os<MODE_, CB_, TP_, false, true>), grid, 256, Lc.stream, p); } else
#define RY_OS_CASE_NU(MODE_) if (!usrc) { RY_LAUNCH((ry_c1d_os<MODE_, 2, 4, false, false>), grid, 256, Lc.stream, p); } else
#define RY_OS_CASE_PM(CB_, TP_) if (n_real > 0 && lp.os_cb == CB_ && lp.os_tp == TP_) { RY_LAUNCH((ry_c1d_os<RY_C1D_S1, CB_, TP_, true, true>), grid, 256, Lc.stream, p); } else
    switch (mode) {
        case RY_C1D_S2:
            RY_OS_CASE_NU(RY_C1D_S2) RY_OS_CASE(RY_C1D_S2, 4, 8) RY_OS_CASE(RY_C1D_S2, 4, 4) RY_OS_CASE(RY_C1D_S2, 2, 8) RY_OS_CASE(RY_C1D_S2, 2, 4)
            return fail(RY_EINVAL, "%s: no ry_c1d_os instantiation for slice %dx%d", l.name, lp.os_cb, lp.os_tp);
            break;
        case RY_C1D_S1:
            if (n_real > 0 && mode != RY_C1D_S1) return fail(RY_EINVAL, "%s: the fused pad needs a stride-1 first layer", l.name);
            RY_OS_CASE_PM(4, 8) RY_OS_CASE_PM(4, 4) RY_OS_CASE_PM(2, 8) RY_OS_CASE_PM(2, 4)
            RY_OS_CASE_NU(RY_C1D_S1) RY_OS_CASE(RY_C1D_S1, 4, 8) RY_OS_CASE(RY_C1D_S1, 4, 4) RY_OS_CASE(RY_C1D_S1, 2, 8) RY_OS_CASE(RY_C1D_S1, 2, 4)
            return fail(RY_EINVAL, "%s: no ry_c1d_os instantiation for slice %dx%d", l.name, lp.os_cb, lp.os_tp);
            break;
        default:
            RY_OS_CASE_NU(RY_C1D_DECONV) RY_OS_CASE(RY_C1D_DECONV, 4, 4) RY_OS_CASE(RY_C1D_DECONV, 2, 8) RY_OS_CASE(RY_C1D_DECONV, 2, 4)
            return fail(RY_EINVAL, "%s: no ry_c1d_os instantiation for slice %dx%d", l.name, lp.os_cb, lp.os_tp);
            break;
    }
#undef RY_OS_CASE
#undef RY_OS_CASE_NU
#undef RY_OS_CASE_PM
    return Lc.end();
}

// The Winograd filters of predictor layer i, built when a plan first takes the layer onto that path (2.25 x the floats of the layer's filters, kept in the
// arena the clones of the predictor share): the direct layout [phase][tap][C][N] holds every filter element once -- read it back, transform, upload.
// Never called under stream capture (plans are built before their first run).
static int ensure_wwin(ry_net* net, int i, const float** out) {
    auto& lazy = net->weights->lazy;
    auto it = lazy.find(i);
    if (it == lazy.end()) {
        const Layer& l = net->layers[i];
        if (!l.wdir || !wino_eligible(l, net->desc.ndim)) return fail(RY_ESTATE, "%s: no Winograd form of this layer", l.name);
        const TapTable t = make_taps(l);
        const int C = l.cin(), N = l.cout;
        std::vector<float> wd((size_t)t.nphases * t.ntaps * C * N);
        RT_TRY(rt::d2h(wd.data(), l.wdir, wd.size() * sizeof(float), net->ctx->stream));
        RT_TRY(rt::stream_sync(net->ctx->stream));
        int where[4][4];
        for (int ph = 0; ph < t.nphases; ++ph)
            for (int tt = 0; tt < t.ntaps; ++tt) where[t.ky[ph][tt]][t.kx[ph][tt]] = ph * t.ntaps + tt;
        std::vector<float> w;
        relayout_wino(l, [&](int n, int c, int ky, int kx) { return (double)wd[((size_t)where[ky][kx] * C + c) * N + n]; }, w);
        float* d = nullptr;
        RY_TRY(upload(*net->weights, net->ctx, w, &d));
        it = lazy.emplace(i, d).first;
    }
    *out = it->second;
    return RY_OK;
}

// rows of the 2-D pixel tiles a layer's launch walks (the dead-row crop and the copied padding rows round to whole tile rows); false: raster tiles
static bool plan_tile_rows(const LayerPlan& lp, int Mh, int Mw, int* th, int* tw_out = nullptr) {
    if (lp.path == PATH_WINO) {
        int tw; wino_tile_hw(lp.wino_cfg, lp.wino_mbw, th, &tw);
        if (tw_out) *tw_out = tw;
        return Mh % *th == 0 && Mw % tw == 0;
    }
    int bm, bn; tile_dims(lp.tile, &bm, &bn);
    for (int tw = 16; tw >= 4; tw >>= 1)
        if (bm % tw == 0 && Mw % tw == 0 && Mh % (bm / tw) == 0) { *th = bm / tw; if (tw_out) *tw_out = tw; return true; }
    return false;
}

// ------------------------------------------------------------------------------------------------
// plan construction
// ------------------------------------------------------------------------------------------------
static int build_plan(ry_net* net, Plan& P) {
    const ry_net_desc& d = net->desc;
    const int nd = d.ndim, B = P.B;
    P.lp.assign(16, LayerPlan());
    int H = nd == 2 ? P.T : 1, W = nd == 2 ? d.width : P.T;
    // spatial sizes per layer
    for (int i = 0; i < 16; ++i) {
        const Layer& l = net->layers[i];
        LayerPlan& lp = P.lp[i];
        int hi, wi;
        if (l.src_a < 0) { hi = H; wi = W; } else { hi = P.lp[l.src_a].Ho; wi = P.lp[l.src_a].Wo; }
        lp.Hi = hi; lp.Wi = wi;
        if (l.deconv) { lp.Ho = nd == 2 ? hi * 2 : 1; lp.Wo = wi * 2; }
        else {
            const int span = l.dil * (l.k - 1) + 1;
            lp.Ho = nd == 2 ? (hi + 2 * l.pad - span) / l.stride + 1 : 1;
            lp.Wo = (wi + 2 * l.pad - span) / l.stride + 1;
        }
        if (lp.Ho < 1 || lp.Wo < 1) return fail(RY_EINVAL, "%s: input %dx%d is too small for this predictor", l.name, hi, wi);
        if (l.src_b >= 0 && (P.lp[l.src_b].Ho != hi || P.lp[l.src_b].Wo != wi))
            return fail(RY_EINVAL, "%s: skip connection is %dx%d but decoder is %dx%d (frames must be a multiple of %d)", l.name,
                        P.lp[l.src_b].Ho, P.lp[l.src_b].Wo, hi, wi, 1 << (d.extensive_layers > 0 ? d.extensive_layers - 1 : 0));
        const double taps = (double)ipow((size_t)l.k, nd);
        const double in_area = (double)B * hi * wi, out_area = (double)B * lp.Ho * lp.Wo;
        lp.flops = 2.0 * l.cin() * l.cout * taps * (l.deconv ? in_area : out_area);
        lp.bytes = 4.0 * ((double)l.cin() * l.cout * taps + in_area * l.cin() + out_area * l.cout);
        if ((double)out_area * l.cout >= 2.0e9 || in_area * l.cin() >= 2.0e9)
            return fail(RY_EINVAL, "%s: activation exceeds 2^31 elements; lower the batch", l.name);
        if (nd == 2 && ((double)out_area * l.cout >= 1.0e9 || in_area * l.cin() >= 1.0e9))
            return fail(RY_EINVAL, "%s: activation exceeds 4 GB (32-bit byte offsets of the implicit GEMM); lower the batch", l.name);
    }
    // buffers
    if (nd == 1) {
        P.s1_os = true;                                   // the output-stationary kernels whenever every layer can take them (generic stride / dilation / GLU layers: the weight-streaming kernels)
        for (int i = 0; i < 16; ++i) if (!c1d_os_capable(net->layers[i]) || !net->layers[i].w1os) P.s1_os = false;
        // the pad of the convert wrapper inside the first layer: a stride-1 first layer whose input channels fit one lane set
        P.s1_padfuse = P.s1_os && P.mode == 1 && c1d_mode(net->layers[0]) == RY_C1D_S1 && net->layers[0].cin() <= 64;
    }
    for (int i = 0; i < 16; ++i) {
        const Layer& l = net->layers[i];
        LayerPlan& lp = P.lp[i];
        const size_t out_elems = (size_t)B * lp.Ho * lp.Wo * l.cout;
        if (nd == 1 && P.s1_os) {
            const int mode = c1d_mode(l);
            lp.os_kt = c1d_os_ktw(l.cin());
            choose_os(l, B, mode == RY_C1D_DECONV ? lp.Wi : lp.Wo, &lp.os_cb, &lp.os_tp);
            if (l.cin_b > 0 && l.cin_a % 64 != 0) { lp.os_cb = 2; lp.os_tp = 4; }      // sources split inside a wave: the per-lane form exists for this slice only
            if (i < 15) RY_TRY(P.arena.alloc(&lp.out, out_elems));          // the last layer stores straight into the caller's block
        } else if (nd == 1) {
            const int mode = c1d_mode(l);
            lp.splits = choose_splits_1d(l, B, mode == RY_C1D_DECONV ? lp.Wi : lp.Wo, mode);
            lp.slab_stride = (long long)out_elems;
            RY_TRY(P.arena.alloc(&lp.raw, out_elems * lp.splits));
        } else {
            RY_TRY(alloc_ztail(net->ctx, P.arena, &lp.out, out_elems));
            if (l.wig) {
                const TapTable t = make_taps(l);
                const int M = B * (l.deconv ? lp.Hi * lp.Wi : lp.Ho * lp.Wo);
                const int nk = t.ntaps * (l.cin() / 32);
                lp.path = PATH_IGEMM; lp.tile = 0; lp.splits = 0; lp.kg = 0;
                { lp.tile = g_force[i][0]; lp.splits = g_force[i][1]; lp.kg = g_force[i][2]; }
                if (lp.tile != 0) {                                  // RY_PLAN: refuse a tile that does not divide the output channels
                    int fbm, fbn; tile_dims(lp.tile, &fbm, &fbn);
                    if (l.cout % fbn != 0) return fail(RY_EINVAL, "RY_PLAN: tile %dx%d does not divide the %d output channels of %s", fbm, fbn, l.cout, l.name);
                    if (net->dtype != 0 && fbm > 128) return fail(RY_EINVAL, "RY_PLAN: no bf16 instantiation of the %dx%d tile (%s)", fbm, fbn, l.name);
                }
                // bf16 mode: a layer runs on bf16 operands when its filters were converted and every producer it reads can
                // write a bf16 copy of its output (the first layer, implicit-GEMM layers and their reduce kernels can)
                // split-bf16 mode: the same, for the layers with enough rows to be bound by the matrix pipe (the weight-streaming
                // layers at the bottom of the U-Net would read 1.5 x the filter bytes: they stay exact fp32)
                const bool x3 = net->dtype == 2;
                bool want16 = x3 ? (l.wigx3 && M >= g_x3_min_m) : (net->dtype == 1 && l.wig16);
                for (int src : {l.src_a, l.src_b}) {
                    if (src < 0) continue;
                    const LayerPlan& sp = P.lp[src];
                    if (sp.path != PATH_FIRST && sp.path != PATH_IGEMM && sp.path != PATH_IGEMM_BF16 && sp.path != PATH_OS2D) want16 = false;   // that producer cannot write a bf16 copy
                }
                if (l.src_a < 0) want16 = false;
                if (want16) {
                    lp.path = PATH_IGEMM_BF16; lp.x3 = x3;
                    choose_igemm(l, M, t.nphases, t.ntaps * ((x3 ? 3 : 1) * l.cin() / 64), &lp.tile, &lp.splits, &lp.kg, x3 ? 2 : 1);
                } else {
                    choose_igemm(l, M, t.nphases, nk, &lp.tile, &lp.splits, &lp.kg);
                }
                // the weight-streaming layers with few rows: output-stationary, one node, no slabs (ry_c2d_os) -- exact fp32 layers only
                if (lp.path == PATH_IGEMM && l.w2os && !(g_os2_forced[i] && g_os2_force[i][0] == 0)) {
                    int c[4] = {0, 0, 0, 0};
                    if (g_os2_forced[i]) for (int q = 0; q < 4; ++q) c[q] = g_os2_force[i][q];
                    const int U = t.ntaps * (l.cin() / 64);
                    double cost = 0.0;
                    if (choose_os2(M, l.cout, t.nphases, U, &c[0], &c[1], &c[2], &c[3], &cost) && (g_os2_forced[i] || cost * U <= (double)g_os2_maxcost)) {
                        lp.path = PATH_OS2D; lp.splits = 1; lp.kg = 1;
                        lp.os2_mt4 = c[0]; lp.os2_nt4 = c[1]; lp.os2_waves = c[2]; lp.os2_depth = c[3];
                    } else if (g_os2_forced[i]) {
                        return fail(RY_EINVAL, "RY_OS2: no output-stationary slice %d:%d:%d:%d for %s", c[0], c[1], c[2], c[3], l.name);
                    }
                }
                // the MFMA-bound k4 s2 p1 layers: Winograd F(2x2, 2x2), 9 / 16 of the matrix-pipe work -- exact-fp32 mode only (RY_WINOGRAD=0: the direct kernels, bit-exact reference)
                if (lp.path == PATH_IGEMM && net->dtype == 0 && g_wino && wino_eligible(l, 2) && !(g_wino_forced[i] && g_wino_force[i][0] == 0) &&
                    !(g_force[i][0] || g_force[i][1] || g_force[i][2])) {                  // (a layer whose direct plan RY_PLAN fixes stays direct)
                    const int Mh = l.deconv ? lp.Hi : lp.Ho, Mw = l.deconv ? lp.Wi : lp.Wo;
                    int c[3] = {0, 0, 0};
                    if (g_wino_forced[i]) { c[0] = g_wino_force[i][0]; c[1] = g_wino_force[i][1]; c[2] = g_wino_force[i][2]; }
                    const int npatches = (l.deconv ? 1 : 4) * (l.cin() / 16);
                    if ((g_wino_forced[i] || M >= g_wino_min_m) && choose_wino(Mh, Mw, l.cout, t.nphases, npatches, B, &c[0], &c[1], &c[2])) {
                        lp.path = PATH_WINO; lp.wino_cfg = c[0]; lp.wino_mbw = c[1]; lp.splits = c[2]; lp.kg = 1; lp.tile = 0;
                        const float* ww = nullptr;
                        RY_TRY(ensure_wwin(net, i, &ww));
                    } else if (g_wino_forced[i]) {
                        return fail(RY_EINVAL, "RY_WINO: no Winograd plan %d:%d:%d for %s", c[0], c[1], c[2], l.name);
                    }
                }
                if (lp.splits > 1) RY_TRY(P.arena.alloc(&lp.slabs, out_elems * lp.splits));
            } else {
                lp.path = PATH_DIRECT; lp.splits = 1;
                if (!l.deconv && l.k == 3 && l.stride == 1 && l.pad == 1) {
                    if (l.src_a < 0 && l.cin() == 1 && l.cout % 4 == 0) lp.path = PATH_FIRST;
                    if (i == 15 && l.cout == 1 && l.cin() % 128 == 0 && l.cin_a % 4 == 0) {
                        lp.path = PATH_LAST;      // exp / edge-pad / crop of SuperResolution.convert fused into the last layer
                        lp.last_rows = lp.Ho;             // convert mode: overwritten with n_frames at enqueue time
                        lp.last_cols = P.mode == 1 ? lp.Wo + 1 : lp.Wo;
                        lp.last_exp = P.mode == 1;
                    }
                }
            }
        }
    }
    // bf16 mode: which copies of each activation are needed (fp32 for fp32 consumers and the caller, bf16 for bf16 consumers)
    if (nd == 2 && net->dtype != 0) {
        const bool x3 = net->dtype == 2;
        std::vector<char> need32(16, 0), need16(16, 0);
        need32[15] = 1;
        for (int i = 0; i < 16; ++i)
            for (int src : {net->layers[i].src_a, net->layers[i].src_b})
                if (src >= 0) (P.lp[i].path == PATH_IGEMM_BF16 ? need16 : need32)[src] = 1;
        for (int i = 0; i < 16; ++i) {
            LayerPlan& lp = P.lp[i];
            lp.w32 = need32[i] || lp.path == PATH_DIRECT || lp.path == PATH_LAST; lp.w16 = need16[i];
            lp.o16x3 = x3;
            if (lp.w16) {
                float* q = nullptr;
                RY_TRY(alloc_ztail(net->ctx, P.arena, &q, ((size_t)B * lp.Ho * lp.Wo * net->layers[i].cout * (x3 ? 2 : 1) + 1) / 2));
                lp.out16 = reinterpret_cast<unsigned short*>(q);
            }
        }
    }
    // staging
    const int cin_user = nd == 1 ? d.in_ch : (P.mode == 1 ? d.width + 1 : d.width);
    const int cout_user = nd == 1 ? d.out_ch : (P.mode == 1 ? d.width + 1 : d.width);
    const int rows_user = P.T;                 // convert mode: any n_frames < T shares this plan
    P.user_in_floats = (size_t)B * rows_user * cin_user;
    P.user_out_floats = (size_t)B * rows_user * cout_user;
    RY_TRY(P.arena.alloc(&P.user_in, P.user_in_floats));
    RY_TRY(P.arena.alloc(&P.user_out, P.user_out_floats));
    if (P.mode == 1) {
        RY_TRY(P.arena.alloc(&P.x_in, (size_t)B * P.T * (nd == 1 ? d.in_ch : d.width)));
    } else {
        P.x_in = nullptr;                      // raw forward reads the caller's block directly (cur_in)
    }
    return RY_OK;
}

static RySrc1d src1d_of(const ry_net* net, const Plan& P, int idx) {
    RySrc1d s;
    memset(&s, 0, sizeof s);
    if (idx == -2) { s.C = 0; s.Craw = 1; s.splits = 1; return s; }
    if (idx == -1) {
        s.raw = P.mode == 1 ? P.x_in : P.cur_in; s.C = net->desc.in_ch; s.Craw = s.C; s.splits = 1; s.act = RY_ACT_NONE;
        return s;
    }
    const Layer& l = net->layers[idx];
    const LayerPlan& lp = P.lp[idx];
    s.raw = lp.raw; s.scale = l.scale; s.shift = l.shift; s.slab_stride = lp.slab_stride;
    s.C = l.act == RY_ACT_GLU ? l.cout / 2 : l.cout; s.Craw = l.cout; s.splits = lp.splits; s.act = l.act;
    return s;
}

// enqueue the whole forward of a plan (wrapper kernels included when mode == 1)
static int enqueue_forward(ry_net* net, Plan& P, Launcher& Lc) {
    const ry_net_desc& d = net->desc;
    const int lo = 0, hi = 16;
    const int nd = d.ndim, B = P.B;
    const float slope = d.lrelu_slope;
    // the fused pad takes the column minimum inside the workgroups that reach the padding: one chain of n_frames / 8 load rounds, worth
    // it while the window is short (measured: 300 frames -3 us, 1000 frames +14 us against the separate ry_pad_min_rows node)
    const bool padfuse_now = nd == 1 && P.s1_padfuse && P.n_frames <= 2048;     // [r5] (the cooperative minimum: one round of loads per 1024 frames; was 512 with the per-lane walk)
    if (P.mode == 1 && !padfuse_now) {
        const int cols_in = nd == 1 ? d.in_ch : d.width + 1;
        const int cols_out = nd == 1 ? d.in_ch : d.width;
        // numpy.pad(mode='minimum') over time (+ log and the dropped last bin for stage 2): column minima and the padded
        // block in one launch
        RyPadRowsParams q;
        q.in = P.cur_in; q.minv = nullptr; q.out = P.x_in;
        q.rows_in = P.n_frames; q.cols_in = cols_in; q.rows_out = P.T; q.cols_out = cols_out; q.take_log = nd == 2;
        q.in_bstride = (long long)P.n_frames * cols_in; q.out_bstride = (long long)P.T * cols_out; q.minv_bstride = cols_in;
        dim3 pg((unsigned)((cols_in + 15) / 16), (unsigned)B);
        RY_TRY(Lc.begin("ry_pad_min_rows", "pad", 0, 4.0 * B * (P.n_frames * cols_in + P.T * cols_out), pg));
        if (P.n_frames > 128) RY_LAUNCH(ry_pad_min_rows<64>, pg, 1024, Lc.stream, q);   // one batch of loads per lane up to 512 frames
        else RY_LAUNCH(ry_pad_min_rows<16>, pg, 256, Lc.stream, q);
        RY_TRY(Lc.end());
    }
    // Stage 2, convert wrapper: the wrapper pads every window to T rows and keeps n_frames of the result (SuperResolution.convert crops);
    // a caller that will itself throw away the first / last frames of the window (ConvertStream.process picks the middle of what it
    // converted) can say so (ry_sr_convert_rows).  The last layer then computes output rows [k0, k1) only, reads rows [k0 - 1, k1 + 1) of
    // decoder c6, and nothing ever reads the other rows.  Walking back through the decoder: correct output rows [a, b) of a k4 s2 p1
    // deconvolution need input rows [floor((a - 1) / 2), floor(b / 2) + 1) (output row 2m takes input rows m - 1 and m, row 2m + 1 rows m
    // and m + 1).  Rows are the outermost axis of the NHWC buffers, so a layer simply runs on a row RANGE of the same buffers
    // (LayerPlan::crop_lo / crop_hi; the rows next to the range read as zero padding, which only reaches rows that are not needed).
    // Every layer demands from its producer exactly the (tile-rounded) rows it reads.  The encoder feeds the bottom of the U-Net and stays whole.
    int crop[16], crop0[16];
    for (int i = 0; i < 16; ++i) crop[i] = crop0[i] = 0;
    int k0 = 0, k1 = P.n_frames;
    if (nd == 2 && P.mode == 1 && P.lp[15].path == PATH_LAST) {
        k0 = P.disc_front < P.n_frames ? P.disc_front : 0;
        k1 = P.n_frames - P.disc_back > k0 ? P.n_frames - P.disc_back : P.n_frames;
        if (k1 <= k0) { k0 = 0; k1 = P.n_frames; }
    }
    if (nd == 2 && P.mode == 1 && g_s2_crop && P.lp[15].path == PATH_LAST && net->layers[15].src_a == 14) {
        int need0 = k0 > 0 ? k0 - 1 : 0, need1 = k1 + 1;             // correct rows [need0, need1) wanted from layer i's output
        for (int i = 14; i >= 8; --i) {
            const Layer& l = net->layers[i];
            const LayerPlan& lp = P.lp[i];
            if (need1 > lp.Ho) need1 = lp.Ho;
            if (need0 <= 0 && need1 >= lp.Ho) break;
            if (lp.path != PATH_IGEMM && lp.path != PATH_IGEMM_BF16 && lp.path != PATH_WINO) break;
            if (l.src_a != i - 1) break;
            int r0, r1;
            if (l.deconv) { r0 = need0 > 0 ? (need0 - 1) / 2 : 0; r1 = need1 / 2 + 1; }
            else if (l.k == 1 && l.stride == 1) { r0 = need0; r1 = need1; }
            else break;
            const int Mw = l.deconv ? lp.Wi : lp.Wo, Mh = l.deconv ? lp.Hi : lp.Ho;
            int th = 1;
            if (plan_tile_rows(lp, Mh, Mw, &th)) { r0 = r0 / th * th; r1 = (r1 + th - 1) / th * th; }      // keep the 2-D pixel tiles of the launch: whole tile rows
            if (r1 > lp.Hi) r1 = lp.Hi;
            if (r0 <= 0 && r1 >= lp.Hi) break;
            // measured at 300 frames (round 2, interleaved A/B on one box): decoder c6 (1536 -> 1216 workgroups, six per CU -> five) 208 -> 177 us, decoder c5
            // (512 -> 416, two per CU) 198 -> 193 us, decoder c4 (256 -> 224, one per CU) 196 -> 194 us: a grid of one workgroup per CU
            // gains nothing by itself, but the CUs it leaves idle go to the window on the other lane (ry_vc_set_lanes): 1.160 -> 1.137 ms
            // per window with two lanes, so it is cropped too (RY_S2_CROP=1 keeps such grids whole)
            int bm = 256, bn = 64;
            if (lp.path == PATH_WINO) { if (lp.wino_cfg == 2) bm = 512; } else tile_dims(lp.tile, &bm, &bn);
            const long wgs = (long)(((long)B * Mh * Mw + bm - 1) / bm) * (l.cout / bn) * (l.deconv ? 4 : 1) * lp.splits;
            if (g_s2_crop >= 2 || wgs > 256) { crop0[i] = r0; crop[i] = r1 - r0; need0 = r0; need1 = r1; }
            else { need0 = 0; need1 = lp.Hi; }                      // this layer runs whole: it reads every row of its producer
        }
    }
    // Stage 2, convert wrapper: rows n_frames .. T - 1 of the padded window are copies of ONE row (the column minima, ry_pad_min_rows), so down the
    // encoder every layer has a stretch of output rows that are equal bit for bit (same operands, same order): identical input rows [a, b] give
    // identical output rows [ceil((a + pad) / stride), floor((b - (k - 1) dil + pad) / stride)] -- at 300 of 384 frames 40 of 192 rows of encoder c1,
    // 19 of 96 of c2.  The implicit GEMM leaves the whole tile rows inside the stretch out of its grid and ry_rep_rows copies the row above them:
    // the MFMA time of those tiles goes to the window on the other lane (RY_S2_HOLE=0 computes them; results are bit-identical either way).
    int hole_lo[16], hole_n[16];
    for (int i = 0; i < 16; ++i) hole_lo[i] = hole_n[i] = 0;
    if (nd == 2 && P.mode == 1 && g_s2_hole && P.n_frames < P.T - 2) {
        int a = P.n_frames, b = P.T - 1;
        for (int i = 0; i < 8; ++i) {
            const Layer& l = net->layers[i];
            const LayerPlan& lp = P.lp[i];
            if (l.deconv || l.src_b >= 0 || l.src_a != i - 1) break;
            const int top = b - (l.k - 1) * l.dil + l.pad;
            if (top < 0) break;
            a = (a + l.pad + l.stride - 1) / l.stride; b = top / l.stride;
            if (b >= lp.Ho) b = lp.Ho - 1;
            if (b - a < 1) break;
            if ((lp.path != PATH_IGEMM && lp.path != PATH_IGEMM_BF16 && lp.path != PATH_WINO) || (lp.splits != 1 && lp.path != PATH_WINO) || crop[i] > 0 || (lp.Wo * l.cout) % 8) continue;
            int th = 1, tw = 0;
            if (!plan_tile_rows(lp, lp.Ho, lp.Wo, &th, &tw) || (lp.path != PATH_WINO && tw != 16)) continue;
            const int r0 = (a + 1 + th - 1) / th * th, r1 = (b + 1) / th * th;      // rows [r0, r1) are whole tile rows and copies of row r0 - 1 >= a
            if (r1 - r0 >= th) { hole_lo[i] = r0; hole_n[i] = r1 - r0; }
        }
    }
    for (int i = lo; i < hi; ++i) {
        const Layer& l = net->layers[i];
        const LayerPlan& lp = P.lp[i];
        if (nd == 1 && P.s1_os) {
            const bool fused_pad = l.src_a < 0 && padfuse_now;
            const float* sa = l.src_a < 0 ? ((P.mode == 1 && !padfuse_now) ? P.x_in : P.cur_in) : P.lp[l.src_a].out;
            const float* sb = l.src_b < 0 ? nullptr : P.lp[l.src_b].out;
            const int keep = (i == 15 && P.mode == 1) ? P.n_frames : lp.Wo;      // the last layer crops to the real frames as it stores
            RY_TRY(launch_c1d_os(Lc, l, lp, B, sa, l.cin_a, sb, l.cin_b, i == 15 ? P.cur_out : lp.out, keep, slope, fused_pad ? P.n_frames : 0));
        } else if (nd == 1) {
            RY_TRY(launch_conv1d(Lc, l, lp, B, src1d_of(net, P, l.src_a), src1d_of(net, P, l.src_b), slope));
        } else {
            const bool in16 = lp.path == PATH_IGEMM_BF16;                     // bf16 consumers read the producers' bf16 copies
            const float* s1 = l.src_a < 0 ? (P.mode == 1 ? P.x_in : P.cur_in)
                                          : in16 ? reinterpret_cast<const float*>(P.lp[l.src_a].out16) : P.lp[l.src_a].out;
            const float* s2 = l.src_b < 0 ? nullptr : in16 ? reinterpret_cast<const float*>(P.lp[l.src_b].out16) : P.lp[l.src_b].out;
            LayerPlan lq = lp;
            if (i == 15 && (P.mode == 0 || lp.path == PATH_LAST)) lq.out = P.cur_out;   // last layer writes the caller's block
            if (i == 15 && P.mode == 1 && lp.path == PATH_LAST) { lq.last_rows = k1 - k0; lq.last_row0 = k0; lq.last_out_rows = P.n_frames; lq.flops = lp.flops * (k1 - k0) / lp.Ho; }
            if (hole_n[i] > 0) { lq.hole_lo = hole_lo[i]; lq.hole_n = hole_n[i]; lq.flops = lp.flops * (lp.Ho - hole_n[i]) / lp.Ho; }
            if (crop[i] > 0) { lq.crop_hi = crop[i]; lq.crop_lo = crop0[i]; lq.flops = lp.flops * crop[i] / lp.Hi; lq.bytes = lp.bytes * crop[i] / lp.Hi; }
            if (lq.path == PATH_WINO) {
                auto it = net->weights->lazy.find(i);
                if (it == net->weights->lazy.end()) return fail(RY_ESTATE, "%s: the Winograd filters of this plan are gone", l.name);
                RY_TRY(launch_wino(Lc, l, lq, it->second, B, s1, l.cin_a, s2, l.cin_b, slope));
            } else {
                RY_TRY(launch_conv2d(Lc, l, lq, B, s1, l.cin_a, s2, l.cin_b, slope));
            }
        }
    }
    if (nd == 1 && P.s1_os) {
        // nothing left to do: decoder c7 wrote the cropped, dense result
    } else if (nd == 1) {
        // decoder c7 keeps raw slabs like every stage-1 layer: sum them here, cropping to the real frames in convert mode
        RyMaterializeParams m;
        m.s = src1d_of(net, P, 15); m.L = P.T; m.keep = P.mode == 1 ? P.n_frames : P.T;
        m.npix = (long long)B * m.keep; m.slope = slope;
        m.out = P.cur_out;
        dim3 mg((unsigned)((m.npix * d.out_ch + 255) / 256));
        RY_TRY(Lc.begin("ry_materialize", "decoder/c7", 0, 4.0 * m.npix * d.out_ch * 2, mg));
        RY_LAUNCH(ry_materialize, mg, 256, Lc.stream, m);
        RY_TRY(Lc.end());
    } else if (P.mode == 1 && P.lp[15].path != PATH_LAST) {
        RySrPostParams q;
        q.y = P.lp[15].out; q.out = P.cur_out; q.rows = P.n_frames; q.cols_in = d.width; q.cols_out = d.width + 1;
        q.y_bstride = (long long)P.T * d.width; q.out_bstride = (long long)P.n_frames * (d.width + 1);
        dim3 pg((unsigned)(((long long)P.n_frames * (d.width + 1) + 255) / 256), (unsigned)B);
        RY_TRY(Lc.begin("ry_sr_post", "post", 0, 8.0 * B * P.n_frames * (d.width + 1), pg));
        RY_LAUNCH(ry_sr_post, pg, 256, Lc.stream, q);
        RY_TRY(Lc.end());
    }
    return RY_OK;
}

// RY_AUTOTUNE=1 (opt-in): measure instead of estimate.  After a stage-2 plan is built, every implicit-GEMM layer is timed on the
// device with its own buffers under a short list of (tile, K groups, external splits) candidates around the planner's pick --
// GEMM + reduce launches, HIP events, the best of `reps` rounds -- and the fastest candidate replaces the pick.  The planner's
// estimate is a model fitted to one window size; the sweeps (profiles/*plansweep*) show it 1-4 % off the per-layer optimum, more at
// window sizes it was not fitted on.  Costs a few hundred launches per plan, once per (batch, frames, mode, dtype).  Split-K
// sums are still taken in a fixed order, so results stay deterministic for a given plan -- but two processes may now pick
// different plans and differ in the last bits, which is why this is not the default.
static int autotune_plan(ry_net* net, Plan& P) {
    ry_ctx* ctx = net->ctx;
    const int B = P.B;
    const float slope = net->desc.lrelu_slope;
    rt::Event e0, e1;
    RT_TRY(rt::event_create(&e0)); RT_TRY(rt::event_create(&e1));
    Launcher Lc{nullptr, ctx, net->stream, nullptr, nullptr};
    int rc = RY_OK;
    for (int i = 0; i < 16 && rc == RY_OK; ++i) {
        LayerPlan& lp = P.lp[i];
        if (lp.path != PATH_IGEMM && lp.path != PATH_IGEMM_BF16) continue;
        if (g_force[i][0] || g_force[i][1] || g_force[i][2]) continue;         // RY_PLAN fixes this layer
        const Layer& l = net->layers[i];
        if (l.src_a < 0) continue;
        const bool b16 = lp.path == PATH_IGEMM_BF16;
        const TapTable t = make_taps(l);
        const int M = B * (l.deconv ? lp.Hi * lp.Wi : lp.Ho * lp.Wo);
        const int nk = t.ntaps * ((b16 && lp.x3 ? 3 : 1) * l.cin() / (b16 ? 64 : 32));
        const size_t out_elems = (size_t)B * lp.Ho * lp.Wo * l.cout;
        const float* s1 = b16 ? reinterpret_cast<const float*>(P.lp[l.src_a].out16) : P.lp[l.src_a].out;
        const float* s2 = l.src_b < 0 ? nullptr : b16 ? reinterpret_cast<const float*>(P.lp[l.src_b].out16) : P.lp[l.src_b].out;
        // candidates: the planner's pick first (ties keep it), then tiles x K groups x splits around it
        struct Cand { int tile, kg, splits; };
        std::vector<Cand> cands;
        cands.push_back({lp.tile, lp.kg, lp.splits});
        std::vector<int> tiles;
        if (l.cout % 128 != 0) tiles = {TILE_128x64};
        else if (M <= 64) tiles = {TILE_32x128, TILE_64x128};
        else tiles = {TILE_128x128, TILE_96x128, TILE_64x128};
        static const int split_list[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32, 48, 64, 96, 128};
        for (int tile : tiles)
            for (int kg = 1; kg <= ((M > 64 && nk >= 16) ? 2 : 1); ++kg)
                for (int sp : split_list) {
                    if (sp * kg > nk || sp > (M <= 64 ? 128 : 32)) continue;
                    int bm, bn; tile_dims(tile, &bm, &bn);
                    const long tiles_n = (long)((M + bm - 1) / bm) * (l.cout / bn) * t.nphases;
                    if (sp > 1 && tiles_n * sp > 4096) continue;                      // more than eight rounds of workgroups: never useful
                    if (sp == 1 && tiles_n < 64) continue;                            // a quarter of the CUs: needs split-K
                    if (tile == lp.tile && kg == lp.kg && sp == lp.splits) continue;
                    cands.push_back({tile, kg, sp});
                }
        if (g_autotune_max > 0 && (int)cands.size() > g_autotune_max) cands.resize(g_autotune_max);
        int max_sp = 1;
        for (const Cand& c : cands) max_sp = c.splits > max_sp ? c.splits : max_sp;
        float* tmp_slabs = nullptr;
        if (max_sp > 1) {
            void* q = nullptr;
            if (rt::dmalloc(&q, out_elems * (size_t)max_sp * sizeof(float)) != 0) { (void)rt::last_error(); continue; }   // no room to tune this layer: keep the pick
            tmp_slabs = (float*)q;
        }
        int best = 0; float best_ms = 1e30f;
        for (size_t c = 0; c < cands.size() && rc == RY_OK; ++c) {
            LayerPlan lq = lp;
            lq.tile = cands[c].tile; lq.kg = cands[c].kg; lq.splits = cands[c].splits; lq.slabs = tmp_slabs;
            float ms_best = 1e30f;
            for (int r = 0; r < 1 + g_autotune_reps && rc == RY_OK; ++r) {             // round 0 warms the instruction cache and the L2
                if (rt::event_record(e0, net->stream) != 0) { rc = fail(RY_EHIP, "autotune: event record failed"); break; }
                rc = launch_conv2d(Lc, l, lq, B, s1, l.cin_a, s2, l.cin_b, slope);
                if (rc != RY_OK) break;
                float ms = 0.f;
                if (rt::event_record(e1, net->stream) != 0 || rt::event_sync(e1) != 0 || rt::event_elapsed(&ms, e0, e1) != 0) { rc = fail(RY_EHIP, "autotune: timing failed"); break; }
                if (r > 0 && ms < ms_best) ms_best = ms;
            }
            if (ms_best < best_ms) { best_ms = ms_best; best = (int)c; }
        }
        if (tmp_slabs) { (void)rt::stream_sync(net->stream); rt::dfree(tmp_slabs); }
        if (rc != RY_OK) break;
        if (g_autotune_pick >= 0) best = g_autotune_pick < (int)cands.size() ? g_autotune_pick : (int)cands.size() - 1;   // tests: exercise the replacement
        const Cand& w = cands[best];
        if (w.splits > 1 && w.splits > lp.splits) rc = P.arena.alloc(&lp.slabs, out_elems * (size_t)w.splits);
        lp.tile = w.tile; lp.kg = w.kg; lp.splits = w.splits;
    }
    rt::event_destroy(e0); rt::event_destroy(e1);
    return rc;
}

static int get_plan(ry_net* net, int B, int T, int mode, int n_frames, Plan** out) {
    if (B < 1 || T < 1) return fail(RY_EINVAL, "batch and frames must be positive (got %d, %d)", B, T);
    auto key = std::make_tuple(B, T, mode, 0);        // convert-mode plans are shared by every n_frames with the same padded length
    auto it = net->plans.find(key);
    if (it == net->plans.end()) {
        if (net->plans.size() >= 16) {                         // bounded cache; queued work may still use the old plans' buffers
            RT_TRY(rt::stream_sync(net->stream));
            net->plans.clear();
        }
        std::unique_ptr<Plan> P(new Plan());
        P->B = B; P->T = T; P->mode = mode; P->n_frames = n_frames;
        RY_TRY(build_plan(net, *P));
        if (g_autotune && net->desc.ndim == 2) RY_TRY(autotune_plan(net, *P));
        it = net->plans.emplace(key, std::move(P)).first;
    }
    it->second->n_frames = n_frames;
    *out = it->second.get();
    return RY_OK;
}

static int run_plan(ry_net* net, Plan& P, const float* x, float* y, int on_device) {
    ry_ctx* ctx = net->ctx;
    RT_TRY(rt::set_device(ctx->device));
    const int rows_now = P.mode == 1 ? P.n_frames : P.T;
    const size_t in_bytes = P.user_in_floats / P.T * rows_now * sizeof(float), out_bytes = P.user_out_floats / P.T * rows_now * sizeof(float);
    // device callers: kernels read / write the caller's buffers directly (no staging copies); the graph is
    // captured once per address pair (Plan::gslots).  host callers: the plan's device staging buffers.
    const float* want_in = on_device ? x : P.user_in;
    float* want_out = on_device ? y : P.user_out;
    P.cur_in = want_in; P.cur_out = want_out;
    if (!on_device) RT_TRY(rt::h2d(P.user_in, x, in_bytes, net->stream));
    Launcher Lc{net, ctx, net->stream, nullptr, nullptr};
#ifndef RY_HOST_EMU
    Plan::GraphSlot* G = nullptr;
    for (Plan::GraphSlot& g : P.gslots) if (g.in == want_in && g.out == want_out) G = &g;
    if (!G) {
        if (P.gslots.size() >= 64) {                      // bounded: evict the least recently used pair (its exec may be in flight: drain first)
            size_t lru = 0;
            for (size_t i = 1; i < P.gslots.size(); ++i) if (P.gslots[i].used < P.gslots[lru].used) lru = i;
            RT_TRY(rt::stream_sync(net->stream));
            if (P.gslots[lru].gexec) hipGraphExecDestroy(P.gslots[lru].gexec);
            P.gslots.erase(P.gslots.begin() + (long)lru);
        }
        P.gslots.push_back(Plan::GraphSlot{want_in, want_out, nullptr, false, -1, -1, 0});
        G = &P.gslots.back();
    }
    G->used = ++P.gclock;
    // the captured graph bakes n_frames into the wrapper kernels: replay only for the same n; a new n runs eagerly once and is
    // captured when it repeats (live windows have a constant n; windows cut by the silence gate vary)
    // (the shape a graph bakes in: the real frames and the frames the caller discards at either end)
    const long long shape = (long long)P.n_frames + ((long long)P.disc_front << 20) + ((long long)P.disc_back << 40);
    if (G->gexec && G->graph_n != shape && G->last_n == shape) {
        RT_TRY(rt::stream_sync(net->stream));             // the exec being replaced may still be running
        hipGraphExecDestroy(G->gexec); G->gexec = nullptr; G->tried = false;
    }
    const bool capture_now = net->use_graph && !G->tried && (P.mode == 0 || G->last_n == shape || G->last_n < 0);
    G->last_n = shape;
    auto capture = [&](hipGraphExec_t* ex) -> int {
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(net->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            int r = enqueue_forward(net, P, Lc);
            hipError_t e = hipStreamEndCapture(net->stream, &graph);
            if (r == RY_OK && e == hipSuccess && graph) {
                if (hipGraphInstantiate(ex, graph, nullptr, nullptr, 0) != hipSuccess) *ex = nullptr;
            }
            if (graph) hipGraphDestroy(graph);
            (void)hipGetLastError();
            if (r != RY_OK) return r;
        }
        return RY_OK;
    };
    if (capture_now) {
        G->tried = true;
        G->graph_n = shape;
        RY_TRY(capture(&G->gexec));
    }
    if (G->gexec && G->graph_n == shape) {
        RT_TRY(hipGraphLaunch(G->gexec, net->stream));
    } else
#endif
    {
        RY_TRY(enqueue_forward(net, P, Lc));
    }
    if (!on_device) {
        RT_TRY(rt::d2h(y, P.user_out, out_bytes, net->stream));
        RT_TRY(rt::stream_sync(net->stream));
    }
    return RY_OK;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* ry_last_error(void) { return g_ry_err.c_str(); }

int ry_device_count(void) {
    int n = 0;
    if (rt::device_count(&n) != 0) return 0;
    return n;
}

// process-wide A/B and diagnostic switches, read when a context is created
// RY_PLAN="layer:tile:splits:kgroups,...": read when a context is created and again at every ry_net_set_dtype (which drops
// the launch plans), so that one process can sweep plans (scripts/gpu_x3_plansweep.py, scripts/gpu_lanesweep.py)
static int read_plan_env() {
    memset(g_force, 0, sizeof(g_force));
    memset(g_os2_force, 0, sizeof(g_os2_force)); memset(g_os2_forced, 0, sizeof(g_os2_forced));
    g_os2_maxcost = 4608; g_os2_min_filter = (size_t)1 << 21;
    if (const char* e = getenv("RY_OS2_MAXCOST")) g_os2_maxcost = atoi(e);
    if (const char* e = getenv("RY_OS2_MINW")) g_os2_min_filter = (size_t)atoll(e);
    memset(g_wino_force, 0, sizeof(g_wino_force)); memset(g_wino_forced, 0, sizeof(g_wino_forced));
    g_wino = 1; g_wino_min_m = 512;
    if (const char* e = getenv("RY_WINOGRAD")) g_wino = atoi(e);
    if (const char* e = getenv("RY_WINO_MINM")) g_wino_min_m = atoi(e);
    if (const char* e = getenv("RY_WINO")) {
        for (const char* q = e; q && *q; q = strchr(q, ',') ? strchr(q, ',') + 1 : nullptr) {
            int i = -1, a = 0, b = 0, c = 0;
            const int got = sscanf(q, "%d:%d:%d:%d", &i, &a, &b, &c);
            if (got >= 2 && i >= 0 && i < 16 && a >= 0 && a <= 2 && b >= 0 && c >= 0) {
                g_wino_forced[i] = true; g_wino_force[i][0] = a; g_wino_force[i][1] = got >= 3 ? b : 0; g_wino_force[i][2] = got >= 4 ? c : 0;
            } else {
                return fail(RY_EINVAL, "RY_WINO: expected layer:cfg[:mbw[:splits]][,...]");
            }
        }
    }
    g_poison = 0;
    if (const char* e = getenv("RY_POISON")) g_poison = atoi(e);
    if (const char* e = getenv("RY_OS2")) {
        for (const char* q = e; q && *q; q = strchr(q, ',') ? strchr(q, ',') + 1 : nullptr) {
            int i = -1, a = 0, b = 0, c = 0, d = 0;
            const int got = sscanf(q, "%d:%d:%d:%d:%d", &i, &a, &b, &c, &d);
            if (got >= 2 && i >= 0 && i < 16 && a >= 0 && b >= 0 && c >= 0 && d >= 0) {
                g_os2_forced[i] = true; g_os2_force[i][0] = a; g_os2_force[i][1] = got >= 3 ? b : 0; g_os2_force[i][2] = got >= 4 ? c : 0; g_os2_force[i][3] = got >= 5 ? d : 0;
            } else {
                return fail(RY_EINVAL, "RY_OS2: expected layer:mt4[:nt4[:waves[:depth]]][,...]");
            }
        }
    }
    if (const char* e = getenv("RY_PLAN")) {
        for (const char* q = e; q && *q; q = strchr(q, ',') ? strchr(q, ',') + 1 : nullptr) {
            int i = -1, t = 0, sp = 0, kg = 0;
            if (sscanf(q, "%d:%d:%d:%d", &i, &t, &sp, &kg) >= 2 && i >= 0 && i < 16 && t >= 0 && t <= TILE_96x128 && t != 2 && sp >= 0 && kg >= 0 && kg <= 2) {
                g_force[i][0] = t; g_force[i][1] = sp; g_force[i][2] = kg;
            } else {
                return fail(RY_EINVAL, "RY_PLAN: expected layer:tile:splits:kgroups[,...]");
            }
        }
    }
    return RY_OK;
}

static int read_env_switches() {
    g_autotune = 0; g_autotune_reps = 3; g_autotune_max = 0; g_autotune_pick = -1;      // (re-read by ry_debug_reload_env: an absent variable means the defaults)
    if (const char* e = getenv("RY_AUTOTUNE")) {                                        // "1[:reps[:max[:pick]]]"
        int on = 0, reps = 3, mx = 0, pick = -1;
        if (sscanf(e, "%d:%d:%d:%d", &on, &reps, &mx, &pick) < 1) return fail(RY_EINVAL, "RY_AUTOTUNE: expected 1[:reps[:max[:pick]]]");
        g_autotune = on; g_autotune_reps = reps > 0 ? reps : 1; g_autotune_max = mx; g_autotune_pick = pick;
    }
    g_x3_min_m = 128; g_s2_crop = 2;
    if (const char* e = getenv("RY_X3_MINM")) g_x3_min_m = atoi(e);
    if (const char* e = getenv("RY_S2_CROP")) g_s2_crop = atoi(e);
    g_s2_hole = 1;
    if (const char* e = getenv("RY_S2_HOLE")) g_s2_hole = atoi(e);
    return read_plan_env();
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

static unsigned short host_f2bf(float f) {
    unsigned u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

static inline float host_bf2f(unsigned short h) { const unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

// Split-bf16 filters of one layer from its fp32 fragment-order blocks (wig layout): the K axis of each source (C channels) becomes
// [W_hi | W_hi | W_lo] (3 C), matching the activations' [x_hi | x_lo | x_hi]: the kernel's plain bf16 contraction over that axis is
// x_hi W_hi + x_lo W_hi + x_hi W_lo.  hi = bf16(w), lo = bf16(w - hi), both RNE.
static void build_wigx3(const Layer& l, const std::vector<float>& w32, std::vector<unsigned short>& out) {
    const TapTable t = make_taps(l);
    const int C = l.cin(), N = l.cout, K3 = 3 * C;
    const size_t outer = (size_t)t.nphases * (N / 64) * t.ntaps;
    out.assign(outer * (size_t)K3 * 64, 0);
    for (size_t o = 0; o < outer; ++o)
        for (int kk = 0; kk < K3; ++kk) {
            int seg, c;
            if (kk < 3 * l.cin_a) { seg = kk / l.cin_a; c = kk % l.cin_a; }
            else { const int k2 = kk - 3 * l.cin_a; seg = k2 / l.cin_b; c = l.cin_a + k2 % l.cin_b; }
            for (int nl = 0; nl < 64; ++nl) {
                const float w = w32[(o * (C / 32) + c / 32) * 2048 + wig_inblock(nl, c % 32)];
                const unsigned short hi = host_f2bf(w);
                out[(o * (K3 / 64) + kk / 64) * 4096 + wig16_inblock(nl, kk % 64)] = seg == 2 ? host_f2bf(w - host_bf2f(hi)) : hi;
            }
        }
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

static int profile_plan(ry_net* net, Plan* P, int reps, ry_kernel_stat* stats, int max_stats, int* n_stats);

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

static int profile_plan(ry_net* net, Plan* P, int reps, ry_kernel_stat* stats, int max_stats, int* n_stats) {
    ry_ctx* ctx = net->ctx;
    RT_TRY(rt::set_device(ctx->device));
    P->cur_in = P->user_in; P->cur_out = P->user_out;        // the plan's own staging: whatever the caller's last blocks were, they may be gone
    std::vector<KernelRec> rec;
    std::vector<double> total;
    for (int r = 0; r < reps; ++r) {
        std::vector<KernelRec> rr;
        std::vector<std::pair<rt::Event, rt::Event>> ev;
        Launcher Lc{net, ctx, net->stream, &rr, &ev};
        int rc = enqueue_forward(net, *P, Lc);
        if (rc == RY_OK && rt::stream_sync(net->stream) != 0) rc = fail(RY_EHIP, "stream sync failed while profiling");
        if (rc == RY_OK) {
            if (total.empty()) total.assign(ev.size(), 0.0);
            for (size_t i = 0; i < ev.size() && i < total.size(); ++i) {
                float ms = 0.f;
                rt::event_elapsed(&ms, ev[i].first, ev[i].second);
                total[i] += ms;
            }
            rec = rr;
        }
        for (auto& pr : ev) { rt::event_destroy(pr.first); rt::event_destroy(pr.second); }
        if (rc != RY_OK) return rc;
    }
    int n = (int)rec.size();
    if (n > max_stats) n = max_stats;
    for (int i = 0; i < n; ++i) {
        memset(&stats[i], 0, sizeof stats[i]);
        snprintf(stats[i].name, sizeof stats[i].name, "%s", rec[i].name.c_str());
        snprintf(stats[i].layer, sizeof stats[i].layer, "%s", rec[i].layer.c_str());
        stats[i].ms = (float)(total[i] / reps);
        stats[i].flops = rec[i].flops; stats[i].bytes = rec[i].bytes; stats[i].flops_exec = rec[i].flops_exec;
        for (int k = 0; k < 3; ++k) stats[i].grid[k] = rec[i].grid[k];
    }
    *n_stats = n;
    return RY_OK;
}


// diagnostics: do two HIP streams of this process really run side by side?  A one-wave kernel that spins for `us` microseconds is put on
// stream i and on stream j; ratio[i * n + j] = wall time of the pair / us: ~1 when the two hardware queues are served together, ~2 when
// one waits for the other (both streams folded onto one queue, or two queues on one pipe of the command processor).
#ifndef RY_HOST_EMU
__global__ void ry_spin_kernel(unsigned long long ticks, unsigned long long* sink) {
    const unsigned long long t0 = wall_clock64();                        // constant-rate counter (hipDeviceAttributeWallClockRate kHz)
    unsigned long long t = t0;
    while (t - t0 < ticks) t = wall_clock64();
    if (sink && t == 1) sink[0] = t;
}
#endif

int ry_debug_stream_overlap(ry_ctx* ctx, int n, int us, float* ratio) {
    if (!ctx || !ratio || n < 2 || n > 32 || us < 10) return fail(RY_EINVAL, "bad argument");
#ifdef RY_HOST_EMU
    for (int i = 0; i < n * n; ++i) ratio[i] = 1.f;
    return RY_OK;
#else
    RT_TRY(rt::set_device(ctx->device));
    std::vector<ry_stream_t> st(n);
    for (int i = 0; i < n; ++i) RT_TRY(rt::stream_create(&st[i]));
    int khz = 100000;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device) != hipSuccess || khz <= 0) khz = 100000;
    const unsigned long long ticks = (unsigned long long)us * (unsigned long long)khz / 1000ull;
    for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(ry_spin_kernel, dim3(1), dim3(64), 0, st[i], 1000ull, nullptr); RT_TRY(rt::stream_sync(st[i])); }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            if (i == j) { ratio[i * n + j] = 1.f; continue; }
            RT_TRY(rt::stream_sync(st[i])); RT_TRY(rt::stream_sync(st[j]));
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(ry_spin_kernel, dim3(1), dim3(64), 0, st[i], ticks, nullptr);
            hipLaunchKernelGGL(ry_spin_kernel, dim3(1), dim3(64), 0, st[j], ticks, nullptr);
            RT_TRY(rt::stream_sync(st[i])); RT_TRY(rt::stream_sync(st[j]));
            const double el = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            ratio[i * n + j] = (float)(el / us);
        }
    for (int i = 0; i < n; ++i) rt::stream_destroy(st[i]);
    return RY_OK;
#endif
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

// ---- single operators -------------------------------------------------------------------------
int ry_conv1d(ry_ctx* ctx, const float* x, int B, int L, int Cin, const float* W, const float* bias, const float* bn,
              int Cout, int k, int stride, int pad, int dilate, int transposed, int act, int splits, float* y) {
    if (!ctx || !x || !W || !y) return fail(RY_EINVAL, "null argument");
    if (B < 1 || L < 1 || Cin < 1 || Cout < 1 || k < 1 || k > 4 || stride < 1 || dilate < 1 || pad < 0)
        return fail(RY_EINVAL, "bad conv1d shape");
    if (transposed && !(k == 4 && stride == 2 && pad == 1 && dilate == 1)) return fail(RY_EINVAL, "transposed conv1d supports k4 s2 p1 only");
    if (act == RY_ACT_GLU && Cout % 2) return fail(RY_EINVAL, "GLU needs an even channel count");
    RT_TRY(rt::set_device(ctx->device));
    Layer l;
    snprintf(l.name, sizeof l.name, "conv1d");
    l.deconv = transposed != 0; l.bn = bn != nullptr; l.k = k; l.stride = stride; l.pad = pad; l.dil = dilate;
    l.cin_a = Cin; l.cout = Cout; l.act = act;
    Arena arena;
    RY_TRY(prepare_layer(ctx, arena, l, 1, 2e-5f, W, bias, bn));
    LayerPlan lp;
    lp.Wi = L;
    lp.Wo = transposed ? 2 * L : (L + 2 * pad - dilate * (k - 1) - 1) / stride + 1;
    if (lp.Wo < 1) return fail(RY_EINVAL, "conv1d output would be empty");
    const int mode = c1d_mode(l);
    lp.splits = splits > 0 ? splits : choose_splits_1d(l, B, transposed ? L : lp.Wo, mode);
    if (lp.splits > Cin) lp.splits = Cin;
    const size_t out_elems = (size_t)B * lp.Wo * Cout;
    lp.slab_stride = (long long)out_elems;
    float *dx = nullptr, *dy = nullptr;
    RY_TRY(arena.alloc(&dx, (size_t)B * L * Cin));
    RY_TRY(arena.alloc(&lp.raw, out_elems * lp.splits));
    const int Cy = act == RY_ACT_GLU ? Cout / 2 : Cout;
    RY_TRY(arena.alloc(&dy, (size_t)B * lp.Wo * Cy));
    RT_TRY(rt::h2d(dx, x, (size_t)B * L * Cin * sizeof(float), ctx->stream));
    Launcher Lc{nullptr, ctx, ctx->stream, nullptr, nullptr};
    RySrc1d sa, sb;
    memset(&sa, 0, sizeof sa); memset(&sb, 0, sizeof sb);
    sa.raw = dx; sa.C = Cin; sa.Craw = Cin; sa.splits = 1; sa.act = RY_ACT_NONE;
    sb.C = 0; sb.Craw = 1; sb.splits = 1;
    RY_TRY(launch_conv1d(Lc, l, lp, B, sa, sb, 0.2f));
    RyMaterializeParams m;
    memset(&m, 0, sizeof m);
    m.s.raw = lp.raw; m.s.scale = l.scale; m.s.shift = l.shift; m.s.slab_stride = lp.slab_stride;
    m.s.C = Cy; m.s.Craw = Cout; m.s.splits = lp.splits; m.s.act = act;
    m.npix = (long long)B * lp.Wo; m.L = lp.Wo; m.keep = lp.Wo; m.out = dy; m.slope = 0.2f;
    dim3 mg((unsigned)((m.npix * Cy + 255) / 256));
    RY_TRY(Lc.begin("ry_materialize", "conv1d", 0, 0, mg));
    RY_LAUNCH(ry_materialize, mg, 256, Lc.stream, m);
    RY_TRY(Lc.end());
    RT_TRY(rt::d2h(y, dy, (size_t)B * lp.Wo * Cy * sizeof(float), ctx->stream));
    RT_TRY(rt::stream_sync(ctx->stream));
    return RY_OK;
}

int ry_conv2d(ry_ctx* ctx, const float* x, int B, int H, int Wd, int Cin, const float* Wt, const float* bias, const float* bn,
              int Cout, int k, int stride, int pad, int transposed, int act, int path, int tile, int splits, float* y) {
    return ry_conv2d_dilated(ctx, x, B, H, Wd, Cin, Wt, bias, bn, Cout, k, stride, pad, 1, transposed, act, path, tile, splits, y);
}

int ry_conv2d_dilated(ry_ctx* ctx, const float* x, int B, int H, int Wd, int Cin, const float* Wt, const float* bias, const float* bn,
                      int Cout, int k, int stride, int pad, int dilate, int transposed, int act, int path, int tile, int splits, float* y) {
    if (!ctx || !x || !Wt || !y) return fail(RY_EINVAL, "null argument");
    if (B < 1 || H < 1 || Wd < 1 || Cin < 1 || Cout < 1 || k < 1 || k > 4 || stride < 1 || pad < 0) return fail(RY_EINVAL, "bad conv2d shape");
    if (dilate < 1 || dilate * (k - 1) > 127) return fail(RY_EINVAL, "dilation %d is out of range", dilate);
    if (dilate != 1 && (transposed || path == PATH_FIRST || path == PATH_LAST)) return fail(RY_EINVAL, "dilation applies to the plain convolution (implicit-GEMM or direct path)");
    if (transposed && !(k == 4 && stride == 2 && pad == 1)) return fail(RY_EINVAL, "transposed conv2d supports k4 s2 p1 only");
    if (act == RY_ACT_GLU) return fail(RY_EINVAL, "GLU is a stage-1 (1-D) epilogue");
    RT_TRY(rt::set_device(ctx->device));
    Layer l;
    snprintf(l.name, sizeof l.name, "conv2d");
    l.deconv = transposed != 0; l.bn = bn != nullptr; l.k = k; l.stride = stride; l.pad = pad; l.dil = dilate;
    l.cin_a = Cin; l.cout = Cout; l.act = act;
    Arena arena;
    RY_TRY(prepare_layer(ctx, arena, l, 2, 2e-5f, Wt, bias, bn, path == PATH_OS2D));
    LayerPlan lp;
    lp.Hi = H; lp.Wi = Wd;
    lp.Ho = transposed ? 2 * H : (H + 2 * pad - dilate * (k - 1) - 1) / stride + 1;
    lp.Wo = transposed ? 2 * Wd : (Wd + 2 * pad - dilate * (k - 1) - 1) / stride + 1;
    if (lp.Ho < 1 || lp.Wo < 1) return fail(RY_EINVAL, "conv2d output would be empty");
    if (path == PATH_IGEMM && !l.wig) return fail(RY_EINVAL, "implicit-GEMM path needs Cin %% 32 == 0 and Cout %% 64 == 0");
    const bool k3 = !transposed && k == 3 && stride == 1 && pad == 1 && dilate == 1;
    if (path == PATH_FIRST && !(k3 && Cin == 1 && Cout % 4 == 0)) return fail(RY_EINVAL, "'first' path is the 1 -> N (N %% 4 == 0) 3x3 layer");
    if (path == PATH_LAST && !(k3 && Cout == 1 && Cin % 128 == 0)) return fail(RY_EINVAL, "'last' path is the C -> 1 (C %% 128 == 0) 3x3 layer");
    if ((path == PATH_IGEMM_BF16 || path == PATH_IGEMM_X3) && !(l.wig && Cin % 64 == 0)) return fail(RY_EINVAL, "bf16 implicit-GEMM path needs Cin %% 64 == 0 and Cout %% 64 == 0");
    lp.path = path ? path : (l.wig ? PATH_IGEMM : PATH_DIRECT);
    if (path == PATH_IGEMM_X3) { lp.path = PATH_IGEMM_BF16; lp.x3 = true; }
    if (path == PATH_WINO) {                 // `tile` = cfg + 16 mbw (zeros: the planner's choice); `splits` external split-K (0: the planner's)
        if (!wino_eligible(l, 2)) return fail(RY_EINVAL, "the Winograd path is the k4 s2 p1 layer with Cin %% 16 == 0 and Cout %% 64 == 0");
        const int Mh = transposed ? H : lp.Ho, Mw = transposed ? Wd : lp.Wo;
        int c[3] = {tile & 15, (tile >> 4) & 15, splits};
        if (!choose_wino(Mh, Mw, Cout, transposed ? 4 : 1, (transposed ? 1 : 4) * (Cin / 16), B, &c[0], &c[1], &c[2]))
            return fail(RY_EINVAL, "no Winograd plan %d:%d:%d for a %d x %d grid", c[0], c[1], c[2], Mh, Mw);
        lp.wino_cfg = c[0]; lp.wino_mbw = c[1]; splits = c[2];
        std::vector<float> w;
        relayout_wino(l, [&](int n, int cc, int ky, int kx) { return (double)w2d_at(l, Wt, n, cc, ky, kx); }, w);
        RY_TRY(upload(arena, ctx, w, &l.wwin));
        tile = 0;
    }
    if (path == PATH_OS2D) {                 // `tile` = mt4 + 16 nt4 + 256 waves + 8192 depth (zeros: the planner's choice)
        if (!l.w2os) return fail(RY_EINVAL, "output-stationary path needs Cin %% 256 == 0 and Cout %% 4 == 0");
        const TapTable t = make_taps(l);
        const int M = B * (transposed ? H * Wd : lp.Ho * lp.Wo);
        int c[4] = {tile & 15, (tile >> 4) & 15, (tile >> 8) & 31, (tile >> 13) & 15};
        if (!choose_os2(M, Cout, t.nphases, t.ntaps * (Cin / 64), &c[0], &c[1], &c[2], &c[3]))
            return fail(RY_EINVAL, "no output-stationary slice %d:%d:%d:%d for this shape", c[0], c[1], c[2], c[3]);
        lp.os2_mt4 = c[0]; lp.os2_nt4 = c[1]; lp.os2_waves = c[2]; lp.os2_depth = c[3];
        tile = 0;
    }
    const size_t out_elems = (size_t)B * lp.Ho * lp.Wo * Cout;
    lp.splits = path == PATH_WINO ? splits : 1;
    lp.last_rows = lp.Ho; lp.last_cols = lp.Wo; lp.last_exp = 0;
    if (path == PATH_WINO && lp.splits > 1) RY_TRY(arena.alloc(&lp.slabs, out_elems * lp.splits));
    if (lp.path == PATH_IGEMM || lp.path == PATH_IGEMM_BF16) {
        const TapTable t = make_taps(l);
        const int M = B * (transposed ? H * Wd : lp.Ho * lp.Wo);
        lp.kg = (tile & 16) ? 2 : ((tile & 32) ? 1 : 0);                    // +16: two K groups per workgroup, +32: one, else automatic
        lp.any_m_patch = true;
        tile &= 15;
        lp.tile = tile; lp.splits = splits;
        if (tile < 0 || tile > TILE_96x128 || tile == 2) return fail(RY_EINVAL, "unknown tile");
        if (tile == TILE_128x64 ? Cout % 64 : (tile != 0 && Cout % 128)) return fail(RY_EINVAL, "tile does not divide Cout");
        const bool op16 = lp.path == PATH_IGEMM_BF16;
        choose_igemm(l, M, t.nphases, t.ntaps * (lp.x3 ? 3 * Cin / 64 : Cin / (op16 ? 64 : 32)), &lp.tile, &lp.splits, &lp.kg, lp.x3 ? 2 : (op16 ? 1 : 0));
        if (lp.x3) {
            std::vector<float> w32;
            relayout_igemm(l, Wt, w32);
            std::vector<unsigned short> wx;
            build_wigx3(l, w32, wx);
            RY_TRY(arena.alloc(&l.wigx3, (wx.size() + 1) / 2));
            RT_TRY(rt::h2d(l.wigx3, wx.data(), wx.size() * sizeof(unsigned short), ctx->stream));
            RT_TRY(rt::stream_sync(ctx->stream));
        } else if (op16) {
            // bf16 filters of this single layer
            const size_t n = (size_t)t.nphases * Cout * t.ntaps * Cin;
            std::vector<float> w32;
            relayout_igemm(l, Wt, w32);
            std::vector<unsigned short> w16(n);
            const size_t outer = (size_t)t.nphases * (Cout / 64) * t.ntaps;
            for (size_t o = 0; o < outer; ++o)
                for (int c = 0; c < Cin; ++c)
                    for (int nl = 0; nl < 64; ++nl)
                        w16[(o * (Cin / 64) + c / 64) * 4096 + wig16_inblock(nl, c % 64)] = host_f2bf(w32[(o * (Cin / 32) + c / 32) * 2048 + wig_inblock(nl, c % 32)]);
            RY_TRY(arena.alloc(&l.wig16, (n + 1) / 2));
            RT_TRY(rt::h2d(l.wig16, w16.data(), n * sizeof(unsigned short), ctx->stream));
            RT_TRY(rt::stream_sync(ctx->stream));
        }
        if (lp.splits > 1) RY_TRY(arena.alloc(&lp.slabs, out_elems * lp.splits));
    }
    float* dx = nullptr;
    RY_TRY(alloc_ztail(ctx, arena, &dx, (size_t)B * H * Wd * Cin));
    RY_TRY(arena.alloc(&lp.out, out_elems));
    Launcher Lc{nullptr, ctx, ctx->stream, nullptr, nullptr};
    if (lp.path == PATH_LAST) {
        // exercise the un-materialised skip concat: the channels are handed over as two half-width sources
        const int Ch = Cin / 2;
        const size_t npix = (size_t)B * H * Wd;
        std::vector<float> ha(npix * Ch), hb(npix * Ch);
        for (size_t q = 0; q < npix; ++q) {
            memcpy(&ha[q * Ch], x + q * Cin, Ch * sizeof(float));
            memcpy(&hb[q * Ch], x + q * Cin + Ch, Ch * sizeof(float));
        }
        float* dx2 = nullptr;
        RY_TRY(arena.alloc(&dx2, npix * Ch));
        RT_TRY(rt::h2d(dx, ha.data(), npix * Ch * sizeof(float), ctx->stream));
        RT_TRY(rt::h2d(dx2, hb.data(), npix * Ch * sizeof(float), ctx->stream));
        RT_TRY(rt::stream_sync(ctx->stream));
        RY_TRY(launch_conv2d(Lc, l, lp, B, dx, Ch, dx2, Ch, 0.2f));
    } else if (lp.path == PATH_IGEMM_BF16) {
        // the bf16 kernel reads bf16 activations (in a predictor the producing layer writes them): round the input here
        const size_t nx = (size_t)B * H * Wd * Cin;
        std::vector<unsigned short> x16(lp.x3 ? 2 * nx : nx);
        if (lp.x3) {                                   // split-bf16 sources: [pixel][hi (Cin) | lo (Cin)]
            for (size_t q = 0; q < nx; ++q) {
                const unsigned short hi = host_f2bf(x[q]);
                const size_t pix = q / Cin, c = q % Cin;
                x16[pix * 2 * Cin + c] = hi; x16[pix * 2 * Cin + Cin + c] = host_f2bf(x[q] - host_bf2f(hi));
            }
        } else {
            for (size_t q = 0; q < nx; ++q) x16[q] = host_f2bf(x[q]);
        }
        RT_TRY(rt::dmemset(dx, 0, (nx + ZTAIL) * sizeof(float), ctx->stream));      // the bf16 data ends half way (split-bf16: at the end): zero tail right behind it
        RT_TRY(rt::h2d(dx, x16.data(), x16.size() * sizeof(unsigned short), ctx->stream));
        RT_TRY(rt::stream_sync(ctx->stream));
        RY_TRY(launch_conv2d(Lc, l, lp, B, dx, Cin, nullptr, 0, 0.2f));
    } else if ((lp.path == PATH_OS2D && Cin % 512 == 0) || (lp.path == PATH_WINO && Cin % 32 == 0)) {
        // exercise the un-materialised skip concat: the channels are handed over as two half-width sources, each followed by its zero pixel
        const int Ch = Cin / 2;
        const size_t npix = (size_t)B * H * Wd;
        std::vector<float> ha(npix * Ch), hb(npix * Ch);
        for (size_t q = 0; q < npix; ++q) {
            memcpy(&ha[q * Ch], x + q * Cin, Ch * sizeof(float));
            memcpy(&hb[q * Ch], x + q * Cin + Ch, Ch * sizeof(float));
        }
        float* dx2 = nullptr;
        RY_TRY(alloc_ztail(ctx, arena, &dx2, npix * Ch));
        RT_TRY(rt::dmemset(dx + npix * Ch, 0, ZTAIL * sizeof(float), ctx->stream));      // the first half ends inside dx: its zero pixel right behind it
        RT_TRY(rt::h2d(dx, ha.data(), npix * Ch * sizeof(float), ctx->stream));
        RT_TRY(rt::h2d(dx2, hb.data(), npix * Ch * sizeof(float), ctx->stream));
        RT_TRY(rt::stream_sync(ctx->stream));
        l.cin_a = Ch; l.cin_b = Ch;
        RY_TRY(launch_conv2d(Lc, l, lp, B, dx, Ch, dx2, Ch, 0.2f));
    } else {
        RT_TRY(rt::h2d(dx, x, (size_t)B * H * Wd * Cin * sizeof(float), ctx->stream));
        RY_TRY(launch_conv2d(Lc, l, lp, B, dx, Cin, nullptr, 0, 0.2f));
    }
    RT_TRY(rt::d2h(y, lp.out, out_elems * sizeof(float), ctx->stream));
    RT_TRY(rt::stream_sync(ctx->stream));
    return RY_OK;
}

}  // extern "C"
