// ry_exec.cpp -- the executor of libry355.so: the one unit that instantiates the gfx950 kernels (ry_kernels.h).  Launchers per kernel family, the whole
// forward of a plan enqueued on the predictor's stream (captured once into a hipGraph, replayed per window), plan autotuning, per-launch profiling, and the
// single operators of the C ABI (ry_conv1d / ry_conv2d: the kernels behind one call, what the operator-level parity tests drive).
#include "ry_kernels.h"
#include "ry_plan.h"

static void fill_geom(RyConvGeom& g, const Layer& l, const LayerPlan& lp, int B, const float* s1, int C1, const float* s2, int C2) {
    memset(&g, 0, sizeof g);
    const TapTable t = make_taps(l);
    g.src1 = s1; g.src2 = s2; g.C1 = C1; g.C2 = C2; g.S1 = C1; g.S2 = C2;
    // K = [hi | lo | hi(wrapped)] over [hi | lo] pixels
    if (lp.path == PATH_IGEMM_BF16 && lp.x3) { g.C1 = 3 * C1; g.C2 = 3 * C2; g.S1 = 2 * C1; g.S2 = 2 * C2; }
    g.B = B; g.Hi = lp.Hi; g.Wi = lp.Wi; g.Ho = lp.Ho; g.Wo = lp.Wo;
    g.Hs = lp.Hi; g.Hos = lp.Ho;
    // a row range of every image in the same buffers; the rows around it read as padding
    if (lp.crop_hi > 0) { g.Hi = lp.crop_hi; g.Ho = l.deconv ? 2 * lp.crop_hi : lp.crop_hi; }
    if (l.deconv) { g.Mh = g.Hi; g.Mw = lp.Wi; g.stride = 1; g.pad = 0; g.ostride = 2; }
    else { g.Mh = g.Ho; g.Mw = lp.Wo; g.stride = l.stride; g.pad = l.pad; g.ostride = 1; }
    g.nphases = t.nphases; g.ntaps = t.ntaps; g.N = l.cout; g.kw = l.deconv ? 2 : l.k; g.dil = l.deconv ? 1 : l.dil;
    const size_t esize = lp.path == PATH_IGEMM_BF16 ? 2 : 4;                  // implicit-GEMM sources end in a zeroed tail (ZTAIL floats)
    g.zoff1 = (unsigned)((size_t)B * lp.Hi * lp.Wi * g.S1 * esize); g.zoff2 = (unsigned)((size_t)B * lp.Hi * lp.Wi * g.S2 * esize);
    // the range starts crop_lo rows into every image: move the bases, keep the zero tails where they are
    if (lp.crop_hi > 0 && lp.crop_lo > 0) {
        const size_t o1 = (size_t)lp.crop_lo * lp.Wi * g.S1 * esize, o2 = (size_t)lp.crop_lo * lp.Wi * g.S2 * esize;
        g.src1 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(s1) + o1); g.zoff1 -= (unsigned)o1;
        if (s2) { g.src2 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(s2) + o2); g.zoff2 -= (unsigned)o2; }
    }
    for (int ph = 0; ph < 4; ++ph) {
        g.pdy[ph] = (signed char)t.pdy[ph]; g.pdx[ph] = (signed char)t.pdx[ph];
        for (int tt = 0; tt < 16; ++tt) { g.tdy[ph][tt] = (signed char)t.dy[ph][tt]; g.tdx[ph][tt] = (signed char)t.dx[ph][tt]; }
    }
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
    // the kernel's pixel offsets (zp1 / zp2, its offset table) are 32-bit byte offsets
    if (((size_t)B * lp.Hi * lp.Wi * (size_t)(C1 > C2 ? C1 : C2) + ZTAIL) * 4 >= ((size_t)1 << 32))
        return fail(RY_EINVAL, "%s: a source of 4 GiB or more does not fit the output-stationary kernel's 32-bit offsets", l.name);
    p.zp1 = (unsigned)((size_t)B * lp.Hi * lp.Wi * C1 * 4); p.zp2 = (unsigned)((size_t)B * lp.Hi * lp.Wi * C2 * 4);
    p.inv_Mimg = 1.f / (float)(p.Mh * p.Mw); p.inv_Mw = 1.f / p.Mw; p.inv_mtiles = 1.f / p.mtiles; p.inv_ntiles = 1.f / p.ntiles; p.inv_cpt = 1.f / cpt;
    p.kw = l.deconv ? 2 : l.k; p.dil = l.deconv ? 1 : l.dil; p.inv_kw = 1.f / p.kw;
    const int total = p.mtiles * p.ntiles * p.nphases;
    dim3 grid((unsigned)(((total + 7) / 8) * 8));
    char nm[48];
    const bool xl = os2_xl_ok(lp.os2_mt4, lp.os2_waves, lp.os2_depth);
    // as rocprofv3 prints it
    snprintf(nm, sizeof nm, "ry_c2d_os<%d,%d,%d,%d,%s>", lp.os2_mt4, lp.os2_nt4, lp.os2_waves, lp.os2_depth, xl ? "true" : "false");
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

// the rows a launch left out of its grid (LayerPlan::hole_*): copies of the row above them, into the fp32 output and / or the bf16 copy ([pixel][N]
// or split [pixel][hi | lo])
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
    // one window: only the rows this launch wrote (a prefix when cropped)
    r.total = B == 1 ? (long long)Ho_run * lp.Wo * l.cout : slab_stride; r.N = l.cout;
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
    // output rows start (2 x for the sub-pixel form) crop_lo rows into every image
    const size_t oo = lp.crop_hi > 0 ? (size_t)(l.deconv ? 2 : 1) * lp.crop_lo * lp.Wo * l.cout : 0;
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
    // (with an external split the rows left out of the grid have no slabs: the reduce node writes whatever their slab memory holds, the copy node
    // behind it fills them in)
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
            // every XCD owns a contiguous band of output rows (45.0 -> 31.5 us against raster order)
            p.xcd_band = 1;
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
    // as rocprofv3 prints it (MODE: 0 = k4 s2 conv, 1 = stride-1 conv, 2 = k4 s2 deconv)
    snprintf(nm, sizeof nm, "ry_c1d_os<%d,%d,%d,%s,%s>", mode, lp.os_cb, lp.os_tp, n_real > 0 ? "true" : "false", usrc ? "true" : "false");
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
    // [r5] (the cooperative minimum: one round of loads per 1024 frames; was 512 with the per-lane walk)
    const bool padfuse_now = nd == 1 && P.s1_padfuse && P.n_frames <= 2048;
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
            // keep the 2-D pixel tiles of the launch: whole tile rows
            if (plan_tile_rows(lp, Mh, Mw, &th)) { r0 = r0 / th * th; r1 = (r1 + th - 1) / th * th; }
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
int autotune_plan(ry_net* net, Plan& P) {
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
            // no room to tune this layer: keep the pick
            if (rt::dmalloc(&q, out_elems * (size_t)max_sp * sizeof(float)) != 0) { (void)rt::last_error(); continue; }
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

int get_plan(ry_net* net, int B, int T, int mode, int n_frames, Plan** out) {
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

int run_plan(ry_net* net, Plan& P, const float* x, float* y, int on_device) {
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

int profile_plan(ry_net* net, Plan* P, int reps, ry_kernel_stat* stats, int max_stats, int* n_stats) {
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

extern "C" int ry_debug_stream_overlap(ry_ctx* ctx, int n, int us, float* ratio) {
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

// ---- single operators -------------------------------------------------------------------------
extern "C" {
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
        // the bf16 data ends half way (split-bf16: at the end): zero tail right behind it
        RT_TRY(rt::dmemset(dx, 0, (nx + ZTAIL) * sizeof(float), ctx->stream));
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
