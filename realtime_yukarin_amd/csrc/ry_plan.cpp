// ry_plan.cpp -- the planner of libry355.so: U-Net topology, filter re-layout at predictor creation (BN folded to scale / shift), the per-layer choice of
// kernel family, tile, split-K and K groups (stage 2: implicit GEMM / Winograd / output-stationary; stage 1: output-stationary slices), the launch plan of a
// (batch, frames) window with its activation buffers, and the environment switches.  No device code: the launchers are ry_exec.cpp.
#include "ry_plan.h"

// ------------------------------------------------------------------------------------------------
// topology (same K-list order as realtime_yukarin_amd/netspec.py)
// ------------------------------------------------------------------------------------------------
static const int ENC_CH[8] = {1, 2, 4, 8, 8, 8, 8, 8};
static const int DEC_IN[7] = {8, 16, 16, 16, 16, 8, 4};
static const int DEC_OUT[7] = {8, 8, 8, 8, 4, 2, 1};

std::vector<Layer> build_topology(const ry_net_desc& d) {
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

size_t ipow(size_t b, int e) { size_t r = 1; while (e-- > 0) r *= b; return r; }

size_t layer_param_count(const Layer& l, int ndim) {
    size_t n = (size_t)l.cin() * l.cout * ipow((size_t)l.k, ndim) + l.cout;
    if (l.bn) n += 4 * (size_t)l.cout;
    return n;
}

int check_desc(const ry_net_desc* d) {
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

TapTable make_taps(const Layer& l) {
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
float w2d_at(const Layer& l, const float* W, int n, int c, int ky, int kx) {
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
size_t wig_inblock(int nl, int k) {
    return (size_t)(nl >> 5) * 1024 + (size_t)(k >> 3) * 256 + (size_t)((((k >> 2) & 1) * 32 + (nl & 31)) * 4) + (size_t)(k & 3);
}

// bf16 blocks of 64 output channels x 64 input channels, same idea: [n/32 : 2][s : 4][lane : 64][j : 8] with
// lane = 32 * lh + (n % 32) and k = 16 s + 8 lh + j (the 8 bf16 a lane feeds to one v_mfma_f32_32x32x16_bf16); index in bf16 units.
size_t wig16_inblock(int nl, int k) {
    return (size_t)(nl >> 5) * 2048 + (size_t)(k >> 4) * 512 + (size_t)((((k >> 3) & 1) * 32 + (nl & 31)) * 8) + (size_t)(k & 7);
}

void relayout_igemm(const Layer& l, const float* W, std::vector<float>& out) {
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
    // (rounds of four 64-channel units inside one source; a source's zero pixel is ZTAIL floats)
    return l.cin_a % 256 == 0 && l.cin_b % 256 == 0 && l.cout % 4 == 0 && l.cin() > 0 && l.cin_a <= 2048 && l.cin_b <= 2048;
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
bool wino_eligible(const Layer& l, int ndim) {
    return ndim == 2 && l.k == 4 && l.stride == 2 && l.pad == 1 && l.dil == 1 && l.cin_a % 16 == 0 && l.cin_b % 16 == 0 && l.cout % 64 == 0 &&
           l.cin_a > 0 && l.cin_a <= 2032 && l.cin_b <= 2032;     // (the channel offset of a patch rides on the base of its zero-tail fetches: ZTAIL floats)
}

// ------------------------------------------------------------------------------------------------
// per-layer launch plans
// ------------------------------------------------------------------------------------------------
// Filters above this size (floats) get the ry_c2d_os layout next to the implicit-GEMM one when a predictor is created: the layers whose time
// is the stream of their filters (SYN-64: encoder c4 .. c7, decoder c0 .. c3, 16.8 - 33.5 MB each).  Whether a plan uses it depends on the
// window (build_plan: few enough output pixels).
static size_t g_os2_min_filter = (size_t)1 << 21;      // RY_OS2_MINW (floats; tests lower it so that small predictors take the path)

int prepare_layer(ry_ctx* ctx, Arena& arena, Layer& l, int ndim, float eps, const float* W, const float* b, const float* bn, bool want_os2) {
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

// RY_POISON=1 (diagnostics): fresh activation buffers are filled with NaN patterns, so that a kernel that reads a row / pixel its producer
// never wrote shows up as NaN in the result instead of depending on what the allocator handed out
static int g_poison = 0;
int alloc_ztail(ry_ctx* ctx, Arena& arena, float** p, size_t nfloats) {
    RY_TRY(arena.alloc(p, nfloats + ZTAIL));
    if (g_poison) RT_TRY(rt::dmemset(*p, 0xFF, nfloats * sizeof(float), ctx->stream));
    RT_TRY(rt::dmemset(*p + nfloats, 0, ZTAIL * sizeof(float), ctx->stream));
    RT_TRY(rt::stream_sync(ctx->stream));
    return RY_OK;
}

void tile_dims(int tile, int* bm, int* bn) {
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
int g_s2_hole = 1;     // RY_S2_HOLE=0: the encoder computes the identical padding rows behind the real frames instead of copying them (A/B, bit-identity tests)
// RY_S2_CROP=0: every decoder layer of the convert wrapper computes all padded rows (A/B of the dead-row crop, used by the bit-identity tests); 1:
// only grids of more than one workgroup per CU
int g_s2_crop = 2;
int g_force[16][3];    // RY_PLAN="layer:tile:splits:kgroups,...": tuning aid, fixes the stage-2 plan of single layers (0 = planner's choice)
// RY_X3_MINM: split-bf16 mode runs a layer on the bf16 pipe from this many GEMM rows (per phase) up (measured at 300 frames: 1 / 32 / 64 / 128 / 512
// / 2048 -> 0.861 / 0.865 / 0.867 / 0.864 vs 0.840 / 0.928 ms per step on two boxes; 128 beat 512 by 1 % in the same-box A/B)
int g_x3_min_m = 128;
// RY_AUTOTUNE="1[:reps[:max[:pick]]]": time candidate launch plans of every stage-2 implicit-GEMM layer on the device when a plan is built
// (autotune_plan)
int g_autotune = 0;
int g_autotune_reps = 3, g_autotune_max = 0;   // ... timed rounds per candidate; cap on the candidates per layer (0 = all; tests)
int g_autotune_pick = -1;                      // ... (tests only) take candidate `pick` of every layer instead of the fastest
// Kernel names as rocprofv3 prints them (template arguments, no spaces): bench.py matches them against profiles/*.
const char* tile_name(int tile, int kg, bool bf16, int patch) {
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

int tile_occ(int tile, int kg) {
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
double est_time(long blocks, int bm, int bn, int s, int occ, int kg, int M, int N, int nk) {
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
        // weight streaming: two workgroups per CU keep enough loads in flight (1024 measured 30 % slower: more slabs, same bandwidth)
        if (tinyM) { const long g = blocks * s; t = g >= 512 ? 1.0 + 1e-4 * s : 512.0 / (double)g; }
        else t = est_time(blocks, bm, bn, s, occ, kg, M, N, nk);
        if (t < bt - 1e-9) { bt = t; best = s; }
    }
    if (t_out) *t_out = bt;
    return best;
}


void choose_igemm(const Layer& l, int M, int nphases, int nk, int* tile, int* splits, int* kg, int bf16) {
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

static bool os2_has_config(int mt4, int nt4, int waves, int depth) {
#define X(A, B, C, D) if (mt4 == A && nt4 == B && waves == C && depth == D) return true;
    RY_OS2_CONFIGS(X)
#undef X
    return false;
}

// RY_OS2_MAXCOST: a layer with the ry_c2d_os filter layout runs output-stationary when slice cost x K units stays below this (0: never).
// Fitted: encoder c6 / decoder c1 at 300 frames (4096) win by 2-4 us, encoder c5 at 300 frames (10240) and
// decoder c2 at 100 frames (8192) lose by 8-11
static int g_os2_maxcost = 4608;
// RY_OS2="layer:mt4:nt4:waves:depth,...": tuning aid, fixes the slice of single layers ("layer:0" keeps that layer on the implicit GEMM)
static int g_os2_force[16][4];
static bool g_os2_forced[16];

// Slice of one layer, by a cost fitted to the slice sweeps on the MI355X (profiles/r05/e_os_sweep_n{300,100}.txt): a workgroup pulls K x (rows +
// channels) of its tile through its CU's L1, the pixel rows at about half the rate of the filter rows (a wave-load of pixels is four
// 256-byte pieces of four different pixels, a wave-load of filters one contiguous KiB), and the launch takes as many rounds as there are
// workgroups per CU.  cost = max(1, workgroups / 256) x (2 rows + channels) of the tile (1.5 rows where the pixels travel by DMA); the sweeps rank
// the slices of every bottom layer in this
// order (encoder c7: 4 x 8 < 8 x 4 < 4 x 4 < 12 x 4; decoder c1: 24 x 16 < 12 x 16 < 8 x 16 < 16 x 16).  `cost_out` x K units is what the
// caller compares with the implicit GEMM (g_os2_maxcost).
bool choose_os2(int M, int N, int nphases, int U, int* mt4, int* nt4, int* waves, int* depth, double* cost_out) {
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
            // a forced pair outside the preference lists (only one of the two forced and no preferred pair fits: no plan)
            if (w == 0 && *waves != 0 && *depth != 0)
                if (U % (4 * *waves) == 0 && os2_has_config(m, n, *waves, *depth)) { w = *waves; d = *depth; }
            if (w == 0) continue;
            const double wgs = (double)((M + 4 * m - 1) / (4 * m)) * (N / (4 * n)) * nphases;
            // pixel rows by DMA through the LDS: contiguous 256-byte pieces (encoder c6: 12 x 8 ahead of 8 x 16)
            const double px = os2_xl_ok(m, w, d) ? 6.0 : 8.0;
            // padded rows are loaded too
            const double cost = (wgs > 256.0 ? wgs / 256.0 : 1.0) * (px * m + 4.0 * n) * ((double)((M + 4 * m - 1) / (4 * m)) * 4 * m / M);
            if (cost < best - 1e-9) { best = cost; bm = m; bn = n; bw = w; bd = d; }
        }
    if (bm == 0) return false;
    *mt4 = bm; *nt4 = bn; *waves = bw; *depth = bd;
    if (cost_out) *cost_out = best;
    return true;
}

// ---- stage-2 layers in Winograd F(2x2, 2x2) form (ry_wino_ldsdma) ----
// Workgroup shapes: cfg 1 = 2 x 2 waves (two M-blocks of 8 x 16 pixels x 64 channels, one 8-channel slice per iteration, 74 KiB of LDS: two per CU),
// cfg 2 = 4 x 2 waves (four M-blocks x 64 channels, two slices per iteration, 146 KiB: one per CU).  mbw = M-blocks per tile row.
static int g_wino = 1;                 // RY_WINOGRAD=0: every layer keeps the direct implicit GEMM (the bit-exact reference of the Winograd form; A/B)
// RY_WINO_MINM: rows (pixels of one phase) from which an eligible layer takes the Winograd form (512 -> 256: encoder c4 / decoder c3 of the 100-frame
// window, 231.8 -> 240.9 k frames/s)
static int g_wino_min_m = 256;
// RY_WINO="layer:cfg:mbw:splits,...": tuning aid, fixes the Winograd plan of single layers ("layer:0" keeps that layer on the direct kernel)
static int g_wino_force[16][3];
static bool g_wino_forced[16];

bool wino_cfg_dims(int cfg, int* wm, int* wn, int* nsl) {
    if (cfg == 1) { *wm = 2; *wn = 2; *nsl = 1; return true; }
    if (cfg == 2) { *wm = 4; *wn = 2; *nsl = 2; return true; }
    return false;
}
void wino_tile_hw(int cfg, int mbw, int* th, int* tw) {              // pixels of the stencil's output grid per M-tile
    int wm = 2, wn = 2, nsl = 1; wino_cfg_dims(cfg, &wm, &wn, &nsl);
    *th = 8 * (wm / mbw); *tw = 16 * mbw;
}
const char* wino_name(int cfg, int mode) {
    static char buf[4][40];
    char* b = buf[(cfg - 1) * 2 + (mode - 1)];
    int wm = 2, wn = 2, nsl = 1; wino_cfg_dims(cfg, &wm, &wn, &nsl);
    snprintf(b, 40, "ry_wino_ldsdma<%d,%d,%d,%d>", wm, wn, nsl, mode);      // as rocprofv3 prints it
    return b;
}

// Plan of one layer: workgroup shape, tile shape (the squarest one that divides the grid: the patch carries one extra row and column), external split-K.
// Returns false when no tile shape divides the Mh x Mw grid.
bool choose_wino(int Mh, int Mw, int N, int nphases, int npatches, int B, int* cfg, int* mbw, int* splits) {
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
            // short tiles: the dead-row crop and the copied padding rows round to whole tile rows (8 x 32 against 16 x 16 tiles, two-lane step at 300
            // frames: encoder -1.1 %, decoder -2.8 %, profiles/r06/plan_ab_mbw.txt)
            const double halo = (double)(th + 1) * (tw + 1) / ((double)th * tw) + 0.006 * th;
            for (int sp = 1; sp <= 32 && sp <= npatches; ++sp) {
                if (*splits != 0 ? *splits != sp : (sp > 1 && sp > npatches / 2)) continue;      // (the planner's own splits leave two patches per workgroup)
                // rounds of workgroups x iterations of the longest split (+ prologue / epilogue of a workgroup, in iterations) x time of an iteration
                // relative to
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

// split heuristic of the weight-streaming stage-1 kernels (128 / 16 measured 0.275 ms, 256 / 32 0.231 ms, 512 / 64 0.238 ms per forward)
static const int g_s1_wgs = 256, g_s1_maxs = 32;

int c1d_mode(const Layer& l) {
    if (l.deconv) return RY_C1D_DECONV;
    if (l.k == 4 && l.stride == 2 && l.pad == 1 && l.dil == 1) return RY_C1D_S2;
    if (l.stride == 1 && l.dil == 1 && l.pad <= 3) return RY_C1D_S1;
    return RY_C1D_GEN;
}

int c1d_tile_len(int mode) { return mode == RY_C1D_DECONV ? 8 : 16; }

int choose_splits_1d(const Layer& l, int B, int rows, int mode) {
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
// smallest workgroup count a layer should reach before it takes a larger slice per workgroup (128 / 256 / 512 / 1024 measured 0.119 / 0.116 / 0.112 /
// 0.116 ms)
static const int g_s1_units = 256;

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
bool plan_tile_rows(const LayerPlan& lp, int Mh, int Mw, int* th, int* tw_out) {
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
int build_plan(ry_net* net, Plan& P) {
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
        // the output-stationary kernels whenever every layer can take them (generic stride / dilation / GLU layers: the weight-streaming kernels)
        P.s1_os = true;
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
            // sources split inside a wave: the per-lane form exists for this slice only
            if (l.cin_b > 0 && l.cin_a % 64 != 0) { lp.os_cb = 2; lp.os_tp = 4; }
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
                    // that producer cannot write a bf16 copy
                    if (sp.path != PATH_FIRST && sp.path != PATH_IGEMM && sp.path != PATH_IGEMM_BF16 && sp.path != PATH_OS2D) want16 = false;
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
                // the MFMA-bound k4 s2 p1 layers: Winograd F(2x2, 2x2), 9 / 16 of the matrix-pipe work -- exact-fp32 mode only (RY_WINOGRAD=0: the
                // direct kernels, bit-exact reference)
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

// process-wide A/B and diagnostic switches, read when a context is created
// RY_PLAN="layer:tile:splits:kgroups,...": read when a context is created and again at every ry_net_set_dtype (which drops
// the launch plans), so that one process can sweep plans (scripts/gpu_x3_plansweep.py, scripts/gpu_lanesweep.py)
int read_plan_env() {
    memset(g_force, 0, sizeof(g_force));
    memset(g_os2_force, 0, sizeof(g_os2_force)); memset(g_os2_forced, 0, sizeof(g_os2_forced));
    g_os2_maxcost = 4608; g_os2_min_filter = (size_t)1 << 21;
    if (const char* e = getenv("RY_OS2_MAXCOST")) g_os2_maxcost = atoi(e);
    if (const char* e = getenv("RY_OS2_MINW")) g_os2_min_filter = (size_t)atoll(e);
    memset(g_wino_force, 0, sizeof(g_wino_force)); memset(g_wino_forced, 0, sizeof(g_wino_forced));
    g_wino = 1; g_wino_min_m = 256;
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

int read_env_switches() {
    // (re-read by ry_debug_reload_env: an absent variable means the defaults)
    g_autotune = 0; g_autotune_reps = 3; g_autotune_max = 0; g_autotune_pick = -1;
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

unsigned short host_f2bf(float f) {
    unsigned u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

float host_bf2f(unsigned short h) { const unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

// Split-bf16 filters of one layer from its fp32 fragment-order blocks (wig layout): the K axis of each source (C channels) becomes
// [W_hi | W_hi | W_lo] (3 C), matching the activations' [x_hi | x_lo | x_hi]: the kernel's plain bf16 contraction over that axis is
// x_hi W_hi + x_lo W_hi + x_hi W_lo.  hi = bf16(w), lo = bf16(w - hi), both RNE.
void build_wigx3(const Layer& l, const std::vector<float>& w32, std::vector<unsigned short>& out) {
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

