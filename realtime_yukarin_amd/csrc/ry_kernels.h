// ry_kernels.h -- hand-written gfx950 (CDNA4) kernels for the convert hot path.
//
// Replaces what Chainer dispatches to cuDNN/CuPy for the two predictors the reference runs per
// buffer (/root/reference/realtime_voice_conversion/yukarin_wrapper/voice_changer.py:33 stage-1,
// :41 stage-2; SURVEY.md section 2.1 kernel inventory).  All activations are channels-last fp32:
//   stage-1  x[b][l][c]        (the (N, C) feature matrix the reference transposes is already this)
//   stage-2  x[b][h][w][c]
// so the GEMM-K axis (input channels) is contiguous and every load/store is a 16-byte lane access.
//
// Kernels:
//   ry_igemm_ldsdma     the stage-2 conv / 4-phase sub-pixel deconv as implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32)
//                       or v_mfma_f32_32x32x16_bf16: operand tiles global -> LDS by DMA with scalar-base addressing, input
//                       patches shared by the taps (PATCH), split-K inside the workgroup (KG), folded BN + activation
//                       epilogue with 16-byte stores, optional split-K slabs; tiles 128x128 / 96x128 / 128x64 / 64x128 / 32x128
//                       (the bf16 form also runs the split-bf16 mode: sources [pixel][hi | lo], K axis [hi | lo | hi] against
//                       filters [w_hi | w_hi | w_lo] = three bf16 products per fp32 product, fp32 accumulate -- DESIGN.md 5.1)
//                       (a hole of whole tile rows can be left out of the grid: the encoder's padding rows that equal the row above, copied by ry_rep_rows)
//   ry_c2d_os           stage-2 layers with a handful of output pixels and megabytes of filters, output-stationary on v_mfma_f32_4x4x1_16B_f32
//                       (sixteen K positions per instruction): one node per layer, no slabs; pixels by LDS-DMA with explicit waits
//   ry_c1d_os           stage-1 layer, output-stationary: lanes over the input channels, every load before the first FMA, DPP reduce-scatter
//   ry_splitk_reduce    sum of split-K slabs + folded BN + activation
//   ry_sr_first / ry_sr_last   the 1 -> N and C -> 1 3x3 end layers of stage 2 (HBM / L2-bound)
//   ry_conv_direct      generic VALU conv (odd channel counts)
//   ry_conv1d_ws        stage-1 weight-streaming 1-D conv/deconv: lanes = output channels (coalesced
//                       16-byte weight reads), LDS-staged input tile whose staging applies the
//                       PRODUCER's split-sum + folded BN + activation (deferred epilogue; skip
//                       connections read through two source descriptors, no materialised concat)
//   ry_materialize      split-sum + folded BN + activation (+GLU) into a dense tensor
//   ry_pad_min_rows / ry_sr_post   the numpy.pad('minimum') / log / exp / edge-pad wrappers
#pragma once
#include "ry_dev.h"


RY_DEV float ry_act(float v, int act, float slope) {
    if (act == RY_ACT_LRELU) return v >= 0.f ? v : v * slope;
    if (act == RY_ACT_RELU) return v > 0.f ? v : 0.f;
    return v;
}
RY_DEV float ry_sigmoid(float v) { return 1.f / (1.f + expf(-v)); }

// Exact x / d for 0 <= x < 2^24, 1 <= d < 2^24: one float multiply by a host-side reciprocal plus a +-1 fix-up (an integer
// division is ~40 instructions on gfx950, and the prologue of the implicit GEMM -- executed once, all workgroups of a
// one-round grid at the same time -- had a dozen of them).
RY_DEV int ry_fdiv(int x, int d, float inv_d) {
    int q = (int)((float)x * inv_d);
    const int r = x - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

// Keeps two values in separate registers: without it the compiler rewrites `c ? v[a] : v[b]` on a register array into a
// select over an INDEX and then emulates the dynamic register index with a chain of A compare / v_cndmask pairs per access
// (measured: the 31-shuffle reduction of 32 sums grew to ~5000 instructions and a layer from 8 to 26-65 us).
RY_DEV void ry_keep2(float& a, float& b) {
#ifndef RY_HOST_EMU
    asm volatile("" : "+v"(a), "+v"(b));
#endif
}
RY_DEV int ry_bitrev(int v, int bits) {            // reverse the low `bits` bits
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}

// four fp32 -> four bf16 (RNE), one 8-byte store
RY_DEV void ry_st4_bf16(unsigned short* q, f32x4 v) {
    u16x4 h;
    h[0] = ry_f2bf(v[0]); h[1] = ry_f2bf(v[1]); h[2] = ry_f2bf(v[2]); h[3] = ry_f2bf(v[3]);
    *reinterpret_cast<u16x4*>(q) = h;
}
RY_DEV float ry_bf2f(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
// Split-bf16 copy of four channels n .. n + 3 of pixel `pix` (N channels): the pixel keeps [hi (N) | lo (N)] with
// hi = bf16(v), lo = bf16(v - hi) (both RNE; v - hi is exact in fp32), so that hi + lo carries 16 mantissa bits of v.
// The implicit GEMM then runs hi*hi + lo*hi + hi*lo on the bf16 matrix pipe with fp32 accumulation (split-bf16).
RY_DEV void ry_st4_bf16_x3(unsigned short* base, size_t pix, int N, int n, f32x4 v) {
    u16x4 h, l;
#pragma unroll
    for (int u = 0; u < 4; ++u) { h[u] = ry_f2bf(v[u]); l[u] = ry_f2bf(v[u] - ry_bf2f(h[u])); }
    unsigned short* q = base + pix * (size_t)(2 * N) + n;
    *reinterpret_cast<u16x4*>(q) = h;
    *reinterpret_cast<u16x4*>(q + N) = l;
}

// ---------------------------------------------------------------------------------------------
// Convolution geometry shared by the implicit-GEMM and the direct kernel.
// GEMM rows enumerate (b, ry, rx) over an Mh x Mw grid per image; input coordinate of tap t is
// (ry*stride - pad + tdy[t], rx*stride - pad + tdx[t]); output pixel is (ry*ostride + pdy, rx*ostride + pdx).
//   conv  k s p : Mh=Ho, Mw=Wo, stride=s, pad=p, ostride=1, 1 phase, taps (ky,kx)
//   deconv k4s2p1: Mh=Hi, Mw=Wi, stride=1, pad=0, ostride=2, 4 phases of 2x2 taps (sub-pixel form)
// ---------------------------------------------------------------------------------------------
struct RyConvGeom {
    const float* src1;
    const float* src2;          // second source of a skip concat (channels C1..C1+C2), or null
    int C1, C2;
    int S1, S2;                 // LDS-DMA kernel: elements per pixel of each source (= C1 / C2, except in the split-bf16 mode where a pixel
                                // keeps [hi | lo] = 2 C bf16 and the K axis runs over [hi | lo | hi] = 3 C: a channel offset past S wraps to 0)
    int B, Hi, Wi, Ho, Wo;
    // rows per image IN MEMORY of the sources / the output (>= Hi / Ho: a launch may cover a row prefix of every image, see LayerPlan::crop_hi)
    int Hs, Hos;
    int Mh, Mw;
    int stride, pad, ostride;
    int nphases, ntaps;
    int kw;                     // taps per kernel row (conv: k, sub-pixel deconv: 2); taps are row-major
    int dil;                    // dilation of a convolution (tap (ky, kx) reads input offset (ky, kx) * dil); 1 for the sub-pixel deconvolution
    int N;                      // output channels
    unsigned zoff1, zoff2;      // LDS-DMA kernel: byte offset of >= 16 zero bytes behind each source (the buffers carry a zeroed tail): padded
                                // rows are fetched from there, so every lane of a piece shares ONE scalar base (scalar-base addressing mode)
    signed char tdy[4][16], tdx[4][16];
    signed char pdy[4], pdx[4];
};

struct RyIgemmParams {
    RyConvGeom g;
    const float* wt;            // [phase][N/64][tap][(C1+C2)/32][64][32]: every (64 couts x 32 k) chunk is one contiguous 8 KB block
    const float* scale;         // [N] folded BN scale (1 when no BN)
    const float* shift;         // [N] folded bias/BN shift
    float* out;                 // splits==1: NHWC output (may be null when only out16 is wanted); else slabs [split][B*Ho*Wo][N] of raw sums
    unsigned short* out16;      // splits==1: optional bf16 copy of the activated output (consumers on the bf16 path), else unused
    int x3;                     // out16 format: 0 = [pixel][N] bf16; 1 = split-bf16 [pixel][hi (N) | lo (N)], lo = bf16(v - hi)
    int splits;
    int act;
    float slope;
    long long slab_stride;
    int mtiles, ntiles;         // 1-D XCD-aware grid: logical id = ((split*mtiles + mt)*ntiles + nt)*nphases + phase
    // host-side helpers of the LDS-DMA kernel's prologue (reciprocals for ry_fdiv, the 2-D tile grid, the K split)
    float inv_nphases, inv_ntiles, inv_mtiles, inv_Mimg, inv_Mw, inv_cpt, inv_kw, inv_tcols, inv_trows;
    int tw_shift, th, tcols, trows;   // 2-D M-tiles: tw = 1 << tw_shift columns x th rows, tcols x trows tiles per image
    // LDS-DMA kernel: XCD grouping (gs slice groups of xcd_nsg slices x 8/gs M-tile groups of xcd_mtg tiles); 0 = contiguous runs
    int xcd_gs, xcd_gs_shift, xcd_nsg, xcd_mtg;
    float inv_xcd_nsg, inv_nsl; // reciprocals of xcd_nsg and of the slice count splits * ntiles * nphases
    int kq, krem;               // K chunks per split: split s takes kq + (s < krem) chunks starting at s * kq + min(s, krem)
    // 2-D M-tiles: tile rows hole_ty .. hole_ty + hole_nt - 1 of every image are not computed (`trows` counts the computed ones): rows of the
    // padding that equal the row above them, filled in by ry_rep_rows (hole_nt = 0: none)
    int hole_ty, hole_nt;
    int tw;                     // > 0: an M-tile is a 2-D block of (BM/tw) x tw rows of the Mh x Mw grid (compact input footprint:
                                //      overlapping taps hit L2); 0: BM consecutive rows in raster order
};

// ---------------------------------------------------------------------------------------------
// ry_igemm_ldsdma<BM, BN, WM, WN, KG, BF16, PATCH> -- the stage-2 implicit GEMM (DESIGN.md 5.1 and section 9 have the measurements behind
// every choice below).
//   BF16 = false: fp32 operands, v_mfma_f32_32x32x2_f32, 32 input channels per K chunk (exact fp32: the headline path);
//   BF16 = true:  bf16 activations and filters, v_mfma_f32_32x32x16_bf16, 64 channels per chunk (BASELINE config #5).
// Operand tiles go global -> LDS by DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction), issued by the same waves that
// run the MFMAs: no staging registers, no ds_write, one barrier per K chunk.  Every piece uses the scalar-base addressing
// mode (wave-uniform base + 32-bit lane offset); padding comes from the zeroed tail behind the source buffer
// (RyConvGeom::zoff1 / zoff2) so that it shares the base.
//   A (pixels x channels): LDS rows of 128 bytes, unpadded (the DMA destination is lane-linear); the 16-byte slot c of row r
//     is stored at position c ^ ((r >> 1) & 7), which makes the ds_read_b128 fragment reads conflict-free (bank = (addr / 4)
//     mod 64 inside the instruction's 16-lane groups).  The swizzle is applied on the SOURCE side: the lane that fills
//     position q of row r fetches slot q ^ f(r).
//     PATCH = 0: every tap gathers its own BM rows.  PATCH = 1 / 2: the taps of a deconvolution phase / of one input parity
//     of a k4 s2 convolution read one shared (BM / 16 + 1) x 17-pixel patch at compile-time row offsets (see below).
//   B (filters): stored in fragment order by the host (ry_plan.cpp: wig_inblock / wig16_inblock), one 1-KiB piece per
//     (32 columns, K step); copied as is, read back lane-linearly.
// Buffers are double, held in DISTINCT __shared__ arrays with loops unrolled so that every buffer index is static: the
// compiler's LDS-DMA alias tracking then does not order the reads of one buffer behind the DMA into the other.
// KG = 2 (512 threads): two groups of four waves take the two halves of the workgroup's K range with their own buffers and
// are summed through the LDS at the end -- split-K without slabs in HBM or a reduce launch.
// Epilogue: folded BN + activation in registers, 32 x 32 tiles transposed through the LDS, 16-byte stores (fp32 and / or a
// bf16 copy for bf16 consumers), or raw split-K slabs.
// ---------------------------------------------------------------------------------------------
template <int V> struct RyConst { static constexpr int value = V; };
// K steps of an iteration over which the DMA pieces of the next one are spread: all four in fp32 (a burst in front of the MFMAs costs 5 %), the
// first one in bf16 (DESIGN.md ledger: 4 / 2 / 1 steps -> 0.824 / 0.787 / 0.776 ms; fragments requested a K step ahead: no gain, round 4)
constexpr int RY_F32_ISSUE_STEPS = 4, RY_BF16_ISSUE_STEPS = 1;

template <int BM, int BN, int WM, int WN, int KG, bool BF16, int PATCH>
RY_KERNEL(256 * KG, 2) void ry_igemm_ldsdma(RyIgemmParams p) {
    constexpr int BK = 32, NS = 4;            // LDS rows of 128 bytes: 32 floats or 64 bf16
    constexpr int CK = BF16 ? 64 : 32;        // input channels per K chunk
    constexpr int ES = BF16 ? 8 : 4;          // elements per 16-byte slot
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    // PATCH = 1 (sub-pixel deconvolution, 16-pixel-wide 2-D tiles): the four taps of a phase read overlapping pixels, so the
    // input patch of the tile ((BM / 16 + 1) x 17 pixels) is fetched ONCE per channel chunk and the A fragments of tap
    // (ky, kx) are read from it at a uniform row offset -- 15 instead of 48 A pieces per chunk for a 96-row tile.
    // PATCH = 2 (k4 s2 p1 convolution): tap (ky, kx) = (2a + r, 2b + c) reads input pixel (2 (oy + a) - 1 + r, 2 (ox + b) - 1 + c),
    // so for a fixed row / column parity (r, c) the four taps (a, b) are a 2 x 2 stride-1 stencil on that parity plane: the
    // same patch scheme with one (BM / 16 + 1) x 17 patch per (channel chunk, parity) -- 60 instead of 192 A pieces per chunk.
    constexpr int PW = 17, PR = (BM / 16 + 1) * PW, PG = (PR + 7) / 8;
    constexpr int AROWS = PATCH != 0 ? PG * 8 : BM;   // LDS rows of one A buffer
    constexpr int AG = PATCH != 0 ? PG : BM / 8;      // 1-KiB DMA pieces of the A tile / patch (8 rows each)
    constexpr int BG = BN / 8;                // 1-KiB DMA pieces of the B tile ((32 columns, K step) each)
    constexpr int AI = (AG + 3) / 4, BI = (BG + 3) / 4, NI = AI + BI;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1 && BM % 32 == 0 && BN % 32 == 0 && BM <= 128, "tile shape");
    static_assert(KG == 1 || KG == 2, "one or two K groups of four waves");
    __shared__ __attribute__((aligned(16))) float As0[KG * AROWS * BK];
    __shared__ __attribute__((aligned(16))) float As1[KG * AROWS * BK];
    __shared__ __attribute__((aligned(16))) float Bs0[KG * BN * BK];
    __shared__ __attribute__((aligned(16))) float Bs1[KG * BN * BK];
    __shared__ int rY[BM], rX[BM], rP[BM], rO[BM];

    const RyConvGeom& g = p.g;
    const int tid = (int)threadIdx.x;
    // Workgroup b runs on XCD b % 8 (one L2 each).  A tile is (M-tile mt, filter slice sl = (split, N-tile, phase)): M-tiles
    // share filters, slices share input pixels.  The host splits the 8 XCDs into gm x gs groups (xcd_gs = gs) so that the L2
    // miss traffic gm * (filter bytes) + gs * (input bytes) is smallest: XCD (xm, xs) owns M-tile block xm and slice block xs.
    const int total_tiles = p.splits * p.mtiles * p.ntiles * p.g.nphases;
    int mt, sl;
    if (p.xcd_gs > 0) {
        const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
        const int xm = xcd >> p.xcd_gs_shift, xs = xcd & (p.xcd_gs - 1);
        const int mtl = ry_fdiv(j, p.xcd_nsg, p.inv_xcd_nsg);
        if (mtl >= p.xcd_mtg) return;
        mt = xm * p.xcd_mtg + mtl;
        sl = xs * p.xcd_nsg + (j - mtl * p.xcd_nsg);
    } else {                                               // no even split: contiguous runs of (mt, slice) pairs per XCD
        const int per_xcd = (total_tiles + 7) >> 3;
        const int lid = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
        if (lid >= total_tiles) return;
        const int nsl = p.splits * p.ntiles * p.g.nphases;
        const int q0 = ry_fdiv(lid, nsl, p.inv_nsl);
        const int msl = lid - q0 * nsl;                     // (mt, slice) with the slice fastest ...
        sl = msl; mt = q0;                                  // ... = the old order for split-free launches
    }
    int q_ = ry_fdiv(sl, p.g.nphases, p.inv_nphases);
    const int phase = sl - q_ * p.g.nphases; sl = q_;
    const int split = ry_fdiv(sl, p.ntiles, p.inv_ntiles);
    const int nt = sl - split * p.ntiles;
    const int m0 = mt * BM;
    const int n0 = nt * BN;
    const int Ctot = g.C1 + g.C2;
    const int Mimg = g.Mh * g.Mw;
    const int M = g.B * Mimg;
    const bool subpix = g.ostride == 2;
    const int pdy = subpix ? (phase >> 1) : 0, pdx = subpix ? (phase & 1) : 0;

    const int lane = tid & 63, wave = ry_uniform((tid >> 6) & 3);
    const int grp = KG > 1 ? ry_uniform(tid >> 8) : 0;      // K group of this wave: the groups split the K range of the workgroup
    // ---- B: element offset of this lane's 16 bytes in the (tap 0, chunk 0) block of each (32 columns, K step) piece ----
    const int c32 = Ctot / CK;                 // K blocks of the filter layout (8 KiB each: 64 x 32 floats or 64 x 64 bf16)
    unsigned boff[BI];
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int gi = 4 * j + wave, gq = gi < BG ? gi : 0;
        const int n = n0 + (gq >> 2) * 32;
        boff[j] = (unsigned)((phase * (g.N >> 6) + (n >> 6)) * (g.ntaps * c32) * 2048 + ((n >> 5) & 1) * 1024 + (gq & 3) * 256 + lane * 4);
    }

    const int cpt = Ctot / CK;
    // K ranges are counted in units of one chunk (PATCH: one patch = four taps, so that every K group starts at tap 0)
    constexpr int KU = PATCH != 0 ? 4 : 1;
    const int kc_begin = split * p.kq + (split < p.krem ? split : p.krem);
    const int wg_units = p.kq + (split < p.krem ? 1 : 0);
    const int g_begin = (kc_begin + (KG > 1 ? (wg_units >> 1) * grp : 0)) * KU;   // this K group's share: the first floor(n / 2) units, the rest
    const int nchunks = (KG > 1 ? (grp ? wg_units - (wg_units >> 1) : (wg_units >> 1)) : wg_units) * KU;
    const int max_chunks = ((wg_units + KG - 1) / KG) * KU;                // barrier count is the same for both groups
    float* const A0 = As0 + grp * (AROWS * BK); float* const A1 = As1 + grp * (AROWS * BK);
    float* const B0 = Bs0 + grp * (BN * BK); float* const B1 = Bs1 + grp * (BN * BK);
    // The filters of the first iteration do not depend on the row tables: request them now (from HBM: the longest latency of
    // the prologue) and set up the A side while they travel.  PATCH kernels only -- there the A set-up does not read the LDS
    // row tables, so no barrier (and its vmcnt(0)) sits between this request and the first one of the main loop.
    if (PATCH != 0 && nchunks > 0) {
        const int pi = g_begin >> 2;
        const int chunk = PATCH == 2 ? pi >> 2 : pi;
        const int tapw = PATCH == 2 ? ((pi >> 1) & 1) * 4 + (pi & 1) : 0;      // tap (a, b) = (0, 0) of the first patch
        const unsigned bd0 = (unsigned)((tapw * c32 + chunk) * 2048);
#pragma unroll
        for (int j = 0; j < BI; ++j) {
            const int gi = 4 * j + wave;
            if (BG % 4 == 0 || gi < BG) ry_glds16_off(p.wt, (boff[j] + bd0) * 4u, B0 + gi * 256);
        }
    }

    for (int r = tid; r < BM; r += 256 * KG) {
        const int m = m0 + r;
        int yb = -(1 << 20), xb = 0, pb = 0, ob = -1;
        bool live = m < M;
        int b = 0, ry = 0, rx = 0;
        if (p.tw > 0) {
            const int trow = ry_fdiv(mt, p.tcols, p.inv_tcols);              // tile row counted over the whole batch
            const int tx = mt - trow * p.tcols;
            b = ry_fdiv(trow, p.trows, p.inv_trows);
            const int tyc = trow - b * p.trows;
            const int ty = tyc + (tyc >= p.hole_ty ? p.hole_nt : 0);
            ry = ty * p.th + (r >> p.tw_shift); rx = tx * p.tw + (r & (p.tw - 1));
            live = b < g.B;
        } else if (live) {
            b = ry_fdiv(m, Mimg, p.inv_Mimg); const int rem = m - b * Mimg;
            ry = ry_fdiv(rem, g.Mw, p.inv_Mw); rx = rem - ry * g.Mw;
        }
        if (live) {
            yb = ry * g.stride - g.pad;
            xb = rx * g.stride - g.pad;
            pb = b * g.Hs * g.Wi;
            ob = (b * g.Hos + ry * g.ostride + pdy) * g.Wo + rx * g.ostride + pdx;
        }
        rY[r] = yb; rX[r] = xb; rP[r] = pb; rO[r] = ob;
    }
    if (PATCH == 0) __syncthreads();           // the gather set-up reads the tables; PATCH kernels need rO only in the epilogue

    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    // ---- A: DMA role of this lane = row drow / position dpos inside each 8-row piece this wave fills ----
    const int drow = lane >> 3, dpos = lane & 7;
    int ayb[AI], axb[AI], aoff1[AI], aoff2[AI];
    if (PATCH != 0) {
        // patch pixel of this lane in each piece it fills: its input coordinates (ayb / axb; far outside the image for rows
        // past the patch or the batch) and its element offsets into the two sources; a fetch adds the uniform parity shift
        // (PATCH = 2) and tests the image borders -- misses are fetched from the zeroed tail of the source
        const int trow = ry_fdiv(mt, p.tcols, p.inv_tcols);
        const int tx = mt - trow * p.tcols;
        const int bimg = ry_fdiv(trow, p.trows, p.inv_trows);
        const int tyc = trow - bimg * p.trows;
        const int ty = tyc + (tyc >= p.hole_ty ? p.hole_nt : 0);
        // input pixel of patch (0, 0): deconvolution: taps reach one pixel up / left of the phase; convolution: the parity-(1, 1)
        // pixel 2 * (first output row / column of the tile)
        const int oy0 = PATCH == 1 ? ty * (BM / 16) + pdy - 1 : 2 * ty * (BM / 16);
        const int ox0 = PATCH == 1 ? tx * 16 + pdx - 1 : 2 * tx * 16;
        constexpr int PS = PATCH == 1 ? 1 : 2;       // input pixels per patch pixel step
#pragma unroll
        for (int j = 0; j < AI; ++j) {
            const int pr = (4 * j + wave) * 8 + drow;
            const int py = ry_fdiv(pr, PW, 1.0f / PW), px = pr - py * PW;
            const int iy = oy0 + PS * py, ix = ox0 + PS * px;
            const bool ok = pr < PR && bimg < g.B;
            const int ce = (dpos ^ ((pr >> 1) & 7)) * ES;
            const int pix = (bimg * g.Hs + iy) * g.Wi + ix;
            ayb[j] = ok ? iy : -(1 << 20); axb[j] = ix;
            aoff1[j] = ok ? pix * g.S1 + ce : 0;
            aoff2[j] = ok ? pix * g.S2 + ce : 0;
        }
    } else {
#pragma unroll
    for (int j = 0; j < AI; ++j) {
        const int gi = 4 * j + wave;
        const int row = (gi < AG ? gi : 0) * 8 + drow;
        const int ce = (dpos ^ ((row >> 1) & 7)) * ES;       // element offset of the slot this lane fetches
        ayb[j] = rY[row]; axb[j] = rX[row];
        const int pixb = rP[row] + ayb[j] * g.Wi + axb[j];
        aoff1[j] = (ayb[j] > -(1 << 19)) ? pixb * g.S1 + ce : 0;
        aoff2[j] = (ayb[j] > -(1 << 19)) ? pixb * g.S2 + ce : 0;
    }
    }
    const int sw = (lr >> 1) & 7;                // swizzle key of every A fragment row this lane reads (tile rows are multiples of 32)
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if constexpr (PATCH == 0) {
        int tap = ry_fdiv(g_begin, cpt, p.inv_cpt);
        int cib = g_begin - tap * cpt;
        int ky = ry_fdiv(tap, g.kw, p.inv_kw), kx = tap - ky * g.kw;

        // state of the chunk being fetched (wave-uniform)
        const float* c_src = nullptr; bool c_first = true; int c_delta = 0, c_dy = 0, c_dx = 0; unsigned c_bdelta = 0;
        auto next_chunk = [&]() {
            const int ci0 = cib * CK;
            c_first = ci0 < g.C1;
            c_src = c_first ? g.src1 : g.src2;
            const int Cs = c_first ? g.S1 : g.S2;
            int cil = c_first ? ci0 : ci0 - g.C1;
            if (cil >= Cs) cil -= Cs;                  // split-bf16 sources: the third K segment reads the hi half again
            c_dy = subpix ? pdy - ky : ky * g.dil; c_dx = subpix ? pdx - kx : kx * g.dil;
            c_delta = (c_dy * g.Wi + c_dx) * Cs + cil;
            c_bdelta = (unsigned)((tap * c32 + cib) * 2048);
            if (++cib == cpt) { cib = 0; ++tap; if (++kx == g.kw) { kx = 0; ++ky; } }
        };
        auto dma_item = [&](int q, float* Ad, float* Bd) {     // q-th DMA instruction of this wave for the chunk being fetched
            if (q < AI) {
                const int j = q, gi = 4 * j + wave;
                if (AG % 4 == 0 || gi < AG) {
                    const int iy = ayb[j] + c_dy, ix = axb[j] + c_dx;
                    const bool ok = (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
                    const unsigned eo = (unsigned)((c_first ? aoff1[j] : aoff2[j]) + c_delta);     // elements (fp32 or bf16)
                    ry_glds16_off(c_src, ok ? eo * (BF16 ? 2u : 4u) : (c_first ? g.zoff1 : g.zoff2), Ad + gi * 256);
                }
            } else {
                const int j = q - AI, gi = 4 * j + wave;
                if (BG % 4 == 0 || gi < BG) ry_glds16_off(p.wt, (boff[j] + c_bdelta) * 4u, Bd + gi * 256);
            }
        };

        // A workgroup with at most two chunks per K group (the weight-streaming layers with M <= 64 rows and split-K in the
        // hundreds) fetches both up front: its time is a chain of memory latencies, not MFMA work.
        const bool pre2 = max_chunks <= 2;
        if (nchunks > 0) {
            next_chunk();
    #pragma unroll
            for (int q = 0; q < NI; ++q) dma_item(q, A0, B0);
            if (pre2 && nchunks > 1) {
                next_chunk();
    #pragma unroll
                for (int q = 0; q < NI; ++q) dma_item(q, A1, B1);
            }
        }
        __syncthreads();

        auto run_chunk = [&](auto bufc, int k) {
            constexpr int BUF = decltype(bufc)::value;
            const float* Ac = BUF ? A1 : A0;
            const float* Bc = BUF ? B1 : B0;
            float* An = BUF ? A0 : A1;
            float* Bn = BUF ? B0 : B1;
            const bool more = (k + 1 < nchunks) && !pre2;
            if (more) next_chunk();
            const float* Ab = Ac + ((wm * TM) * 32 + lr) * BK;
            if (KG == 1 || k < nchunks) {
    #pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int pos = ((2 * s + lh) ^ sw) * 4;
                f32x4 af[TM], bf[TN];
    #pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = ry_ld4(Ab + i * 32 * BK + pos);
    #pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = ry_ld4(Bc + ((wn * TN + j) * 4 + s) * 256 + lane * 4);
                constexpr int ISG = BF16 ? NS : RY_F32_ISSUE_STEPS;
                if (more && s < ISG) {
    #pragma unroll
                    for (int q = (s * NI) / ISG; q < ((s + 1) * NI) / ISG; ++q) dma_item(q, An, Bn);
                }
                if (BF16) {                        // one v_mfma_f32_32x32x16_bf16 per fragment pair: the 16 bytes are 8 bf16 of k = 16 s + 8 (lane >> 5) + j
    #pragma unroll
                    for (int i = 0; i < TM; ++i)
    #pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = ry_mfma_32x32x16_bf16(__builtin_bit_cast(u16x8, af[i]), __builtin_bit_cast(u16x8, bf[j]), acc[i][j]);
                } else {
    #pragma unroll
                for (int t = 0; t < 4; ++t)
    #pragma unroll
                    for (int i = 0; i < TM; ++i)
    #pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = ry_mfma_32x32x2(af[i][t], bf[j][t], acc[i][j]);
                }
            }
            }
            __syncthreads();                       // DMA of chunk k + 1 landed (vmcnt) and buffer BUF is free again
        };
        for (int k = 0; k < max_chunks; k += 2) {
            run_chunk(RyConst<0>(), k);
            if (k + 1 < max_chunks) run_chunk(RyConst<1>(), k + 1);
        }

    } else
    {
        // ---------------- PATCH main loop: iteration k = (patch k / 4, tap k % 4 of that patch) ----------------
        // a patch is a channel chunk (deconvolution) or a (channel chunk, input parity) pair (convolution)
        const int patch0 = g_begin >> 2;              // K groups start at tap 0 of a patch
        int pch = patch0;                             // next patch to fetch
        int bit = 0;                                  // next iteration whose filters are fetched
        const float* c_src = nullptr; bool c_first = true; int c_cil = 0, c_py = 0, c_px = 0, c_pdelta = 0; unsigned c_bdelta = 0;
        auto next_patch = [&]() {
            const int chunk = PATCH == 2 ? pch >> 2 : pch;
            const int ci0 = chunk * CK;
            c_first = ci0 < g.C1;
            c_src = c_first ? g.src1 : g.src2;
            c_cil = c_first ? ci0 : ci0 - g.C1;
            if (c_cil >= (c_first ? g.S1 : g.S2)) c_cil -= c_first ? g.S1 : g.S2;      // split-bf16 sources, as in the gather variant
            if (PATCH == 2) {                         // parity (r, c) = ((pch >> 1) & 1, pch & 1): pixel = parity-(1, 1) pixel - (1 - r, 1 - c)
                c_py = ((pch >> 1) & 1) - 1; c_px = (pch & 1) - 1;
                c_pdelta = (c_py * g.Wi + c_px) * (c_first ? g.S1 : g.S2);
            }
            ++pch;
        };
        auto next_b = [&]() {
            const int pi = patch0 + (bit >> 2), ti = bit & 3;
            int tapw, chunk;
            if (PATCH == 2) { chunk = pi >> 2; tapw = (2 * (ti >> 1) + ((pi >> 1) & 1)) * 4 + 2 * (ti & 1) + (pi & 1); }   // (ky, kx) = (2a + r, 2b + c)
            else { chunk = pi; tapw = ti; }
            c_bdelta = (unsigned)((tapw * c32 + chunk) * 2048);
            ++bit;
        };
        auto patch_item = [&](int j, float* Ad) {
            const int gi = 4 * j + wave;
            if (AG % 4 == 0 || gi < AG) {
                const int iy = ayb[j] + c_py, ix = axb[j] + c_px;
                const bool ok = (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
                const unsigned eo = (unsigned)((c_first ? aoff1[j] : aoff2[j]) + c_pdelta + c_cil);
                ry_glds16_off(c_src, ok ? eo * (BF16 ? 2u : 4u) : (c_first ? g.zoff1 : g.zoff2), Ad + gi * 256);
            }
        };
        auto b_item = [&](int j, float* Bd) {
            const int gi = 4 * j + wave;
            if (BG % 4 == 0 || gi < BG) ry_glds16_off(p.wt, (boff[j] + c_bdelta) * 4u, Bd + gi * 256);
        };
        int pbase[TM];                                // patch row of this lane's fragment rows for the tap at offset (0, 0)
#pragma unroll
        for (int i = 0; i < TM; ++i) { const int ml = (wm * TM + i) * 32 + lr; pbase[i] = (ml >> 4) * PW + (ml & 15); }
        if (nchunks > 0) {
            next_patch();
#pragma unroll
            for (int j = 0; j < AI; ++j) patch_item(j, A0);
            next_b();                                 // its pieces were requested at the top of the kernel
        }
        __syncthreads();
        auto run_it = [&](auto k8c, int k) {
            constexpr int K8 = decltype(k8c)::value;
            constexpr int TAP = K8 & 3, ABUF = (K8 >> 2) & 1, BBUF = K8 & 1;
            // patch row offset of the tap: deconvolution (dy - dymin, dx - dxmin) with dy = pdy - ky; convolution (a, b)
            constexpr int TAPOFF = PATCH == 1 ? (1 - (TAP >> 1)) * PW + (1 - (TAP & 1)) : (TAP >> 1) * PW + (TAP & 1);
            const float* Ac = ABUF ? A1 : A0;
            const float* Bc = BBUF ? B1 : B0;
            float* An = ABUF ? A0 : A1;
            float* Bn = BBUF ? B0 : B1;
            const bool more_b = k + 1 < nchunks;
            const bool more_a = TAP == 0 && (k + 4 < nchunks);
            if (more_b) next_b();
            if (more_a) next_patch();
            if (KG == 1 || k < nchunks) {
            // DMA pieces of the next iteration: fp32 spreads them over the NS K steps of this one (a burst in front of the MFMAs
            // costs 5 %); a bf16 iteration is 16 x shorter on the matrix pipe, the pieces have to land before its closing
            // barrier, so they all go out in its FIRST K step (measured, split-bf16 forward: 4 / 2 / 1 steps -> 0.824 / 0.787 / 0.776 ms)
            constexpr int ISL = BF16 ? RY_BF16_ISSUE_STEPS : RY_F32_ISSUE_STEPS;
            auto issue = [&](int s) {
                if (more_b && s < ISL) {
#pragma unroll
                    for (int q = (s * BI) / ISL; q < ((s + 1) * BI) / ISL; ++q) b_item(q, Bn);
                }
                if (TAP == 0 && more_a && s < ISL) {
#pragma unroll
                    for (int q = (s * AI) / ISL; q < ((s + 1) * AI) / ISL; ++q) patch_item(q, An);
                }
            };
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                f32x4 af[TM], bf[TN];
                {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int pr = pbase[i] + TAPOFF;
                    af[i] = ry_ld4(Ac + pr * BK + (((2 * s + lh) ^ ((pr >> 1) & 7)) << 2));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = ry_ld4(Bc + ((wn * TN + j) * 4 + s) * 256 + lane * 4);
                }
                issue(s);
                if (BF16) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = ry_mfma_32x32x16_bf16(__builtin_bit_cast(u16x8, af[i]), __builtin_bit_cast(u16x8, bf[j]), acc[i][j]);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j) acc[i][j] = ry_mfma_32x32x2(af[i][t], bf[j][t], acc[i][j]);
                }
            }
            }
            __syncthreads();
        };
        for (int k = 0; k < max_chunks; k += 8) {
            run_it(RyConst<0>(), k);
            if (k + 1 < max_chunks) run_it(RyConst<1>(), k + 1);
            if (k + 2 < max_chunks) run_it(RyConst<2>(), k + 2);
            if (k + 3 < max_chunks) run_it(RyConst<3>(), k + 3);
            if (k + 4 < max_chunks) run_it(RyConst<4>(), k + 4);
            if (k + 5 < max_chunks) run_it(RyConst<5>(), k + 5);
            if (k + 6 < max_chunks) run_it(RyConst<6>(), k + 6);
            if (k + 7 < max_chunks) run_it(RyConst<7>(), k + 7);
        }
    }

    if (KG > 1) {
        // Sum the two K groups inside the workgroup: group 1 parks its accumulators in the (now idle) B buffers, lane-linear,
        // group 0 adds them in a fixed order (group 0 + group 1) and runs the epilogue.  No slabs, no second kernel.
        float* scr = (wave < 2 ? Bs0 : Bs1) + (wave & 1) * (TM * TN * 16 * 64) + lane;
        static_assert(KG == 1 || 2 * TM * TN * 16 * 64 <= KG * BN * BK, "accumulators of two waves fit one B buffer");
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) scr[((i * TN + j) * 16 + r) * 64] = acc[i][j][r];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += scr[((i * TN + j) * 16 + r) * 64];
        ry_wave_sync();                        // the epilogue reuses this wave's region as its transposition scratch
    }

    // ---- epilogue: folded BN + activation in registers, then every 32 x 32 accumulator tile is transposed through a
    // 4-KiB per-wave LDS scratch (the operand buffers are idle now) so that a lane holds 4 consecutive channels of one
    // pixel: 4 x 16-byte stores per tile instead of 16 x 4-byte ones, one row lookup per store.  (The scalar-store
    // epilogue was ~60 instructions per store and ~10 us of a 125 us layer: all workgroups reach it together.)
    float* outp = p.out + (p.splits > 1 ? (size_t)split * (size_t)p.slab_stride : (size_t)0);
    float* T = (wave < 2 ? Bs0 : Bs1) + (wave & 1) * (KG > 1 ? TM * TN * 16 * 64 : 1024);
    static_assert(BN * BK >= 2048, "two waves' transposition scratch fits one B buffer");
    const bool final_ = p.splits == 1;
    const int erow = lane >> 3, eslot = lane & 7;            // store role: row erow + 8 q of the tile, channels 4 eslot .. + 3
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nbase = n0 + (wn * TN + j) * 32;
        float sc = 1.f, sh = 0.f;
        if (final_) { sc = p.scale[nbase + lr]; sh = p.shift[nbase + lr]; }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
            if (final_) {
                if (p.act == RY_ACT_LRELU) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const float x = fmaf(v[r], sc, sh); v[r] = x >= 0.f ? x : x * p.slope; }
                } else if (p.act == RY_ACT_RELU) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const float x = fmaf(v[r], sc, sh); v[r] = x > 0.f ? x : 0.f; }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = fmaf(v[r], sc, sh);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                T[row * 32 + ((((lr >> 2) ^ (row & 7)) << 2) | (lr & 3))] = v[r];
            }
            ry_wave_sync();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = erow + 8 * q;
                const f32x4 o = ry_ld4(T + row * 32 + ((eslot ^ (row & 7)) << 2));
                const int ob = rO[(wm * TM + i) * 32 + row];
                if (ob >= 0) {
                    const size_t oi = (size_t)ob * g.N + nbase + eslot * 4;
                    if (!final_ || p.out) ry_st4(outp + oi, o);
                    if (final_ && p.out16) {
                        if (p.x3) ry_st4_bf16_x3(p.out16, (size_t)ob, g.N, nbase + eslot * 4, o);
                        else ry_st4_bf16(p.out16 + oi, o);
                    }
                }
            }
            ry_wave_sync();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// ry_wino_ldsdma<WM, WN, NSL, MODE> -- the MFMA-bound stage-2 layers in Winograd F(2x2, 2x2) form (round 6).
// Every k4 s2 p1 layer of the U-Net is a sum of 2 x 2-tap stride-1 stencils: a transposed convolution is four sub-pixel phases of one
// each (MODE 1: output (2 ry + pdy, 2 rx + pdx) = sum_{a,b} in[ry + pdy - 1 + a][rx + pdx - 1 + b] g_phase[a][b]), a convolution is four
// input-parity planes P_rc[y][x] = in[2 y - 1 + r][2 x - 1 + c] summed (MODE 2: out[oy][ox] = sum_{r,c} sum_{a,b} P_rc[oy + a][ox + b] W[2 a + r][2 b + c]).
// F(2x2, 2x2) computes a 2 x 2 block of such a stencil's outputs from its 3 x 3 inputs with 9 products instead of 16:
//   V = B^T d B   (rows, then columns: [d0 - d1, d1, d2 - d1]),   U = G g G^T ([g0, g0 + g1, g1], in float64 on the host, rounded once),
//   M_p = sum over channels (and parities) of V_p U_p for the nine positions p,   Y = A^T M A ([m0 + m1, m1 + m2]).
// -44 % of the matrix-pipe time of the direct form for fp32-class results (measured 3e-7 against 1e-7 of the direct kernel, float64 reference).
// Mapping: an M-block is 4 x 8 Winograd tiles (8 x 16 pixels of the stencil's output grid) = the 32 rows of ONE v_mfma_f32_32x32x2_f32 per
// position; a wave owns one M-block x 32 output channels x 9 positions = 144 accumulator registers, all nine blocks of a lane hold the same
// (tile, channel), so the output transform is per-lane adds.  A workgroup is WM x WN waves: (WM / mbw) x mbw M-blocks by 32 WN channels.
// K loop, one iteration = NSL slices of 8 input channels, all nine positions:
//   A: the raw input patch ((2 TTH + 1) x (2 TTW + 1) pixels, 16 channels = two slices) goes global -> LDS by DMA, double-buffered; a lane
//      reads the 3 x 3 pixels of its tile (nine ds_read_b128 = 4 channels each) and transforms them in registers (12 float4 subtractions).
//      Patch pixel (py, px) sits at position py * PW + (even columns first, then odd) with its four 16-byte slots XOR-ed by (py >> 1) & 3:
//      the nine fragment reads are bank-conflict free (the lanes of a tile row read consecutive positions).
//   B: the transformed filters, stored by the host as [phase][N / 32 WN][slice][position][n / 32][lane][4] = one contiguous 9 WN KiB run per
//      iteration and slice in fragment order, double-buffered.
// One barrier per iteration.  Epilogue: output transform, folded BN + activation, 32 x 32 transposition through the LDS, 16-byte stores
// (or raw split-K slabs for ry_splitk_reduce).  Tile rows left out of the grid (hole_*) and row ranges (Hs / Hos) as in ry_igemm_ldsdma.
// ---------------------------------------------------------------------------------------------
struct RyWinoParams {
    RyConvGeom g;               // src1 / src2, C1 / C2 (S1 = C1, S2 = C2), B, Hi, Wi, Hs, Ho, Wo, Hos, Mh, Mw, ostride, nphases, N, zoff1 / zoff2, pdy / pdx
    const float* wt;            // transformed filters (relayout_wino)
    const float* scale;
    const float* shift;
    float* out;                 // splits == 1: NHWC output; else slabs [split][B * Ho * Wo][N] of raw sums
    int splits;
    int act;
    float slope;
    long long slab_stride;
    int mtiles, ntiles;         // 1-D XCD-aware grid as ry_igemm_ldsdma: logical id = ((split * mtiles + mt) * ntiles + nt) * nphases + phase
    int mbw;                    // M-blocks per tile row: a tile is (WM / mbw) x mbw blocks of 8 x 16 pixels
    int tcols, trows;           // M-tiles per row of the grid, computed tile rows per image
    float inv_nphases, inv_ntiles, inv_tcols, inv_trows, inv_pw;
    int xcd_gs, xcd_gs_shift, xcd_nsg, xcd_mtg;
    float inv_xcd_nsg, inv_nsl;
    int npatches;               // K axis in patches of 16 channels (MODE 2: x 4 parities, parity fastest)
    int kq, krem;               // patches per split: split s takes kq + (s < krem) patches starting at s * kq + min(s, krem)
    int hole_ty, hole_nt;       // as RyIgemmParams
};

template <int WM, int WN, int NSL, int MODE>
RY_KERNEL(64 * WM * WN, 2) void ry_wino_ldsdma(RyWinoParams p) {
    constexpr int NW = WM * WN, NT = 64 * NW;
    // patch positions per A buffer: 17 x 17 / 9 x 33 (WM = 2), 17 x 33 / 33 x 17 / 9 x 65 (WM = 4), rounded up to whole DMA pieces
    constexpr int NPOS = WM == 2 ? 304 : 592;
    constexpr int AG = NPOS / 16;                      // 1-KiB DMA pieces per patch (16 positions x 64 bytes)
    constexpr int AI = (AG + NW - 1) / NW;
    // pieces of the next patch issued in the first iteration of the current one: all of them (NSL = 1: they have the second iteration to land)
    constexpr int AH = AI;
    // patch pieces EVERY wave issues behind its filter pieces (the waves with a piece more wait for their first one too)
    constexpr int AFLY = AG / NW;
    constexpr int BSL = 9 * WN * 256;                  // floats of one filter slice (8 channels x 9 positions x 32 WN output channels)
    constexpr int BG = NSL * 9 * WN;                   // DMA pieces per iteration
    constexpr int BI = (BG + NW - 1) / NW;
    static_assert((WM == 2 || WM == 4) && (WN == 2 || WN == 4) && (NSL == 1 || NSL == 2) && (MODE == 1 || MODE == 2), "shape");
    static_assert(NW * 1024 <= NSL * BSL, "the epilogue's transposition scratch fits one B buffer");
    __shared__ __attribute__((aligned(16))) float As0[NPOS * 16];
    __shared__ __attribute__((aligned(16))) float As1[NPOS * 16];
    __shared__ __attribute__((aligned(16))) float Bs0[NSL * BSL];
    __shared__ __attribute__((aligned(16))) float Bs1[NSL * BSL];
    __shared__ int rO[WM * 32];

    const RyConvGeom& g = p.g;
    const int tid = (int)threadIdx.x;
    const int total_tiles = p.splits * p.mtiles * p.ntiles * g.nphases;
    int mt, sl;
    if (p.xcd_gs > 0) {                                    // XCD (xm, xs) owns M-tile block xm and slice block xs (see ry_igemm_ldsdma)
        const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
        const int xm = xcd >> p.xcd_gs_shift, xs = xcd & (p.xcd_gs - 1);
        const int mtl = ry_fdiv(j, p.xcd_nsg, p.inv_xcd_nsg);
        if (mtl >= p.xcd_mtg) return;
        mt = xm * p.xcd_mtg + mtl;
        sl = xs * p.xcd_nsg + (j - mtl * p.xcd_nsg);
    } else {
        const int per_xcd = (total_tiles + 7) >> 3;
        const int lid = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
        if (lid >= total_tiles) return;
        const int nsl = p.splits * p.ntiles * g.nphases;
        mt = ry_fdiv(lid, nsl, p.inv_nsl);
        sl = lid - mt * nsl;
    }
    int q_ = ry_fdiv(sl, g.nphases, p.inv_nphases);
    const int phase = sl - q_ * g.nphases; sl = q_;
    const int split = ry_fdiv(sl, p.ntiles, p.inv_ntiles);
    const int nt = sl - split * p.ntiles;
    const int n0 = nt * (32 * WN);
    const int pdy = MODE == 1 ? (phase >> 1) : 0, pdx = MODE == 1 ? (phase & 1) : 0;

    const int lane = tid & 63, wave = ry_uniform(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int mbw = p.mbw;                                  // blocks per tile row
    const int TTH = 4 * (WM / mbw), TTW = 8 * mbw;          // Winograd tiles per M-tile
    const int PW = 2 * TTW + 1, PH = 2 * TTH + 1, NE = TTW + 1;   // patch size in pixels; even columns per patch row

    // K range of this workgroup, in patches and in iterations
    const int pbeg = split * p.kq + (split < p.krem ? split : p.krem);
    const int npat = p.kq + (split < p.krem ? 1 : 0);
    const int nit = npat * (2 / NSL);
    // filters of this (phase, N-tile): consecutive slices are consecutive runs of BSL floats
    const float* wt_it = p.wt + ((size_t)(phase * p.ntiles + nt) * (size_t)(2 * p.npatches) + (size_t)(2 * pbeg)) * BSL;
    auto b_item = [&](int j, float* Bd) {
        const int gi = j * NW + wave;
        // (uniform base + 32-bit lane offset: the scalar-base addressing mode)
        if (BG % NW == 0 || gi < BG) ry_glds16_off(wt_it, (unsigned)(gi * 1024 + lane * 16), Bd + gi * 256);
    };
    if (nit > 0) {                                          // the filters of the first iteration travel while the A side is set up
#pragma unroll
        for (int j = 0; j < BI; ++j) b_item(j, Bs0);
        wt_it += NSL * BSL;
    }

    // ---- the M-tile ----
    const int trow = ry_fdiv(mt, p.tcols, p.inv_tcols);
    const int txt = mt - trow * p.tcols;
    const int bimg = ry_fdiv(trow, p.trows, p.inv_trows);
    const int tyc = trow - bimg * p.trows;
    const int tyt = tyc + (tyc >= p.hole_ty ? p.hole_nt : 0);
    const int ry0 = tyt * (2 * TTH), rx0 = txt * (2 * TTW);          // first row / column of the tile on the stencil's output grid
    if (tid < WM * 32) {                                    // output pixel of (a, b) = (0, 0) of every Winograd tile (epilogue)
        const int blk = tid >> 5, lr_ = tid & 31;
        const int tyl = (blk / mbw) * 4 + (lr_ >> 3), txl = (blk % mbw) * 8 + (lr_ & 7);
        const int ry = ry0 + 2 * tyl, rx = rx0 + 2 * txl;
        rO[tid] = bimg < g.B ? (bimg * g.Hos + ry * g.ostride + pdy) * g.Wo + rx * g.ostride + pdx : -1;
    }
    // ---- A: DMA role of this lane in each piece its wave fills: position pp = 16 gi + lane / 4, physical slot lane & 3 ----
    constexpr int PS = MODE == 1 ? 1 : 2;                   // input pixels per patch pixel
    const int oy0 = MODE == 1 ? ry0 + pdy - 1 : 2 * ry0;    // input pixel of patch (0, 0): MODE 2: of the parity-(1, 1) plane
    const int ox0 = MODE == 1 ? rx0 + pdx - 1 : 2 * rx0;
    unsigned aoff1[AI], aoff2[AI];
    unsigned amask = 0;                                     // MODE 2: bit 4 j + 2 r + c = piece j's pixel of parity (r, c) lies inside the image
#pragma unroll
    for (int j = 0; j < AI; ++j) {
        const int gi = j * NW + wave;
        const int pp = gi * 16 + (lane >> 2), ps = lane & 3;
        const int py = ry_fdiv(pp, PW, p.inv_pw), rem = pp - py * PW;
        const int px = rem < NE ? 2 * rem : 2 * (rem - NE) + 1;
        const int ls = ps ^ ((py >> 1) & 3);                // logical slot (4 channels) stored at this physical slot
        const int iy = oy0 + PS * py, ix = ox0 + PS * px;
        const bool inp = gi < AG && py < PH && bimg < g.B;
        const unsigned pix = (unsigned)((bimg * g.Hs + iy) * g.Wi + ix);     // (wraps for pixels outside the image: never used then)
        if (MODE == 1) {
            const bool ok = inp && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
            aoff1[j] = ok ? pix * (unsigned)g.S1 * 4u + (unsigned)ls * 16u : g.zoff1;
            aoff2[j] = ok ? pix * (unsigned)g.S2 * 4u + (unsigned)ls * 16u : g.zoff2;
        } else {
            aoff1[j] = pix * (unsigned)g.S1 * 4u + (unsigned)ls * 16u; aoff2[j] = pix * (unsigned)g.S2 * 4u + (unsigned)ls * 16u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int y2 = iy + (q >> 1) - 1, x2 = ix + (q & 1) - 1;
                if (inp && (unsigned)y2 < (unsigned)g.Hi && (unsigned)x2 < (unsigned)g.Wi) amask |= 1u << (4 * j + q);
            }
        }
    }
    // state of the patch being fetched (wave-uniform)
    int pch = pbeg;
    const float* c_src = nullptr; bool c_first = true; int c_par = 0; unsigned c_delta1 = 0, c_delta2 = 0;
    auto next_patch = [&]() {
        const int chunk = MODE == 2 ? pch >> 2 : pch;
        const int ci0 = chunk * 16;
        c_first = ci0 < g.C1;
        // the channel offset rides on the scalar base (the zero tail is a whole zeroed pixel)
        c_src = (c_first ? g.src1 : g.src2) + (c_first ? ci0 : ci0 - g.C1);
        if (MODE == 2) {
            c_par = pch & 3;
            const int dpix = (((pch >> 1) & 1) - 1) * g.Wi + ((pch & 1) - 1);
            c_delta1 = (unsigned)(dpix * g.S1 * 4); c_delta2 = (unsigned)(dpix * g.S2 * 4);
        }
        ++pch;
    };
    auto a_item = [&](int j, float* Ad) {
        const int gi = j * NW + wave;
        if (AG % NW == 0 || gi < AG) {
            unsigned off;
            if (MODE == 1) off = c_first ? aoff1[j] : aoff2[j];
            else {
                const bool ok = (amask >> (4 * j + c_par)) & 1u;
                off = ok ? (c_first ? aoff1[j] + c_delta1 : aoff2[j] + c_delta2) : (c_first ? g.zoff1 : g.zoff2);
            }
            ry_glds16_off(c_src, off, Ad + gi * 256);
        }
    };
    if (nit > 0) {
        next_patch();
#pragma unroll
        for (int j = 0; j < AI; ++j) a_item(j, As0);
    }
    // ---- A fragments: float index of the lane's 3 x 3 pixels (slice 0; slice 1 = index ^ 8) ----
    const int lr = lane & 31, lh = lane >> 5;
    const int tyl = (wm / mbw) * 4 + (lr >> 3), txl = (wm % mbw) * 8 + (lr & 7);
    int aaddr[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int py = 2 * tyl + i;
            const int rem = j == 1 ? NE + txl : txl + (j >> 1);
            aaddr[i * 3 + j] = (py * PW + rem) * 16 + ((lh ^ ((py >> 1) & 3)) << 2);
        }
    f32x16 acc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    __syncthreads();

    // ONE code path per unrolled iteration (the requests of the next iteration sit behind wave-uniform branches -- issuing them unconditionally, the
    // last iteration
    // re-fetching its own operands, measured +3 % on the convolution layers and +-0 on the others, profiles/r06/dma_uncond_ab.txt): a variant per
    // request pattern as a
    // compile-time argument made hipcc 7.2 keep the 144 accumulator registers of the variants in different places and spill them at the joins.
    auto run_it = [&](auto k4c, int k) {
        constexpr int K4 = decltype(k4c)::value;
        constexpr int SL = NSL == 1 ? (K4 & 1) : 0;                       // slice of the patch this iteration starts with
        constexpr int ABUF = NSL == 1 ? ((K4 >> 1) & 1) : (K4 & 1), BBUF = K4 & 1;
        const float* Ac = ABUF ? As1 : As0;
        const float* Bc = BBUF ? Bs1 : Bs0;
        float* An = ABUF ? As0 : As1;
        float* Bn = BBUF ? Bs0 : Bs1;
        const bool more_b = k + 1 < nit;
        const bool more_a = k - SL + 2 / NSL < nit;                       // a next patch exists
        if (more_a && SL == 0) next_patch();
        // DMA pieces of the next iteration (filters) and of the next patch (its first AH pieces with the first slice, the rest with the second), one
        // per position
        constexpr int A_LO = SL == 0 ? 0 : AH, A_HI = SL == 0 ? AH : AI, NITEM = BI + (A_HI - A_LO);
        auto issue1 = [&](int item) {
            if (item < BI) { if (more_b) b_item(item, Bn); }
            else if (item < NITEM) { if (more_a) a_item(A_LO + item - BI, An); }
        };
        // in order -- filters first (needed at the next barrier), then the patch --, spread evenly over the position steps
        auto issue = [&](int step) {
#pragma unroll
            for (int item = (step * NITEM) / (9 * NSL); item < ((step + 1) * NITEM) / (9 * NSL); ++item) issue1(item);
        };
#pragma unroll
        for (int s = 0; s < NSL; ++s) {
            f32x4 v[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                v[q] = ry_ld4(Ac + (aaddr[q] ^ ((SL + s) * 8)));
            }
            // MFMA order: the three positions of a row TOGETHER, K step by K step -- consecutive MFMAs go to different accumulators.  (The first form
            // ran the four
            // K steps of one position back to back: every VALU / LDS / DMA instruction the scheduler placed between two MFMAs on the SAME accumulator cost
            // ~43 cycles (MI355X_MICROARCH.md, cycle constants) -- the input transform alone 13 % of a launch, profiles/r06/wino_ablation.txt.)  The
            // filter fragments
            // of the next row are requested before the MFMAs of this one.
            auto ldb = [&](int q) -> f32x4 {
                return ry_ld4(Bc + s * BSL + (q * WN + wn) * 256 + lane * 4);
            };
            f32x4 bfa[3], bfb[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) bfa[j] = ldb(j);
            // V = B^T d B: rows, then columns
#pragma unroll
            for (int j = 0; j < 3; ++j) { v[j] -= v[3 + j]; v[6 + j] -= v[3 + j]; }
#pragma unroll
            for (int i = 0; i < 3; ++i) { v[3 * i] -= v[3 * i + 1]; v[3 * i + 2] -= v[3 * i + 1]; }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f32x4 (&bc)[3] = (i & 1) ? bfb : bfa;
                f32x4 (&bn)[3] = (i & 1) ? bfa : bfb;
                if (i + 1 < 3) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) bn[j] = ldb(3 * (i + 1) + j);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int q = 3 * i + j;
                        if (j == 0 && t < 3) issue(s * 9 + 3 * i + t);      // the DMA requests: one step in front of each of the first three K steps of a row
                        acc[q] = ry_mfma_32x32x2(v[q][t], bc[j][t], acc[q]);
                    }
                }
            }
        }
        static_assert(NITEM <= 18 * NSL, "at most two DMA pieces per position step");
        if (more_b) wt_it += NSL * BSL;
        // The filters of iteration k + 1 have to be in the LDS behind this barrier; the pieces of the next patch, requested BEHIND them, are read two barriers
        // from here (NSL = 1) and may stay in flight: vmcnt counts this wave's requests in order, AFLY of the youngest are patch pieces in every wave.
        // (A __syncthreads() here drains every request: the landing time of the youngest -- activations out of another XCD's L2 -- was exposed at
        // every barrier.)
        if (NSL == 1 && SL == 0 && more_a) ry_own_dma_landed<AFLY>(); else ry_own_dma_landed<0>();
        ry_lds_barrier();      // this wave's fragment reads are done (lgkmcnt), every wave's filters landed: the buffers of iteration k are free again
    };
    for (int k = 0; k < nit; k += 4) {
        run_it(RyConst<0>(), k);
        if (k + 1 < nit) run_it(RyConst<1>(), k + 1);
        if (k + 2 < nit) run_it(RyConst<2>(), k + 2);
        if (k + 3 < nit) run_it(RyConst<3>(), k + 3);
    }

    // ---- epilogue: Y = A^T M A per lane, folded BN + activation, 32 x 32 transposition through a 4-KiB per-wave scratch, 16-byte stores ----
    float* outp = p.out + (p.splits > 1 ? (size_t)split * (size_t)p.slab_stride : (size_t)0);
    float* T = Bs0 + wave * 1024;
    const bool final_ = p.splits == 1;
    const int erow = lane >> 3, eslot = lane & 7;
    const int nbase = n0 + wn * 32;
    float sc = 1.f, sh = 0.f;
    if (final_) { sc = p.scale[nbase + lr]; sh = p.shift[nbase + lr]; }
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) {
        const int a = ab >> 1, b = ab & 1;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // (m[a][b] + m[a][b + 1]) + (m[a + 1][b] + m[a + 1][b + 1]), fixed order
            const float top = acc[3 * a + b][r] + acc[3 * a + b + 1][r];
            const float bot = acc[3 * a + 3 + b][r] + acc[3 * a + 3 + b + 1][r];
            v[r] = top + bot;
        }
        if (final_) {
            if (p.act == RY_ACT_LRELU) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float x = fmaf(v[r], sc, sh); v[r] = x >= 0.f ? x : x * p.slope; }
            } else if (p.act == RY_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float x = fmaf(v[r], sc, sh); v[r] = x > 0.f ? x : 0.f; }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaf(v[r], sc, sh);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            T[row * 32 + ((((lr >> 2) ^ (row & 7)) << 2) | (lr & 3))] = v[r];
        }
        ry_wave_sync();
        const int dpix = (a * g.Wo + b) * g.ostride;             // output pixel of (a, b) relative to (0, 0) of the tile
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = erow + 8 * q;
            const f32x4 o = ry_ld4(T + row * 32 + ((eslot ^ (row & 7)) << 2));
            const int ob = rO[wm * 32 + row];
            if (ob >= 0) ry_st4(outp + (size_t)(ob + dpix) * g.N + nbase + eslot * 4, o);
        }
        ry_wave_sync();
    }
}

struct RyReduceParams {
    const float* slabs;
    int splits;
    long long slab_stride;
    const float* scale;
    const float* shift;
    float* out;                 // fp32 output, or null
    unsigned short* out16;      // bf16 copy of the output (consumers on the bf16 path), or null
    int x3;                     // out16 format: 0 = plain bf16, 1 = split-bf16 [pixel][hi | lo] (ry_st4_bf16_x3)
    long long total;            // elements (multiple of 4)
    int N;
    int act;
    float slope;
};

RY_KERNEL(256) void ry_splitk_reduce(RyReduceParams p) {
    const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= p.total) return;
    // four independent partial sums (slabs k = 0,1,2,3 mod 4) keep >= 4 loads in flight; the order is fixed,
    // so the result is deterministic
    const int n = (int)((unsigned)i4 % (unsigned)p.N);          // (the executor keeps every activation below 2^31 elements: 32-bit arithmetic)
    const f32x4 sc = ry_ld4(p.scale + n), sh = ry_ld4(p.shift + n);       // requested with the slabs, not behind them
    f32x4 s0 = ry_ld4(p.slabs + i4), s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1, s3 = s1;
    const size_t st = (size_t)p.slab_stride;
    int k = 1;
    for (; k + 3 < p.splits; k += 4) {
        const f32x4 a = ry_ld4(p.slabs + (size_t)k * st + i4), b = ry_ld4(p.slabs + (size_t)(k + 1) * st + i4);
        const f32x4 c = ry_ld4(p.slabs + (size_t)(k + 2) * st + i4), d = ry_ld4(p.slabs + (size_t)(k + 3) * st + i4);
        s1 += a; s2 += b; s3 += c; s0 += d;
    }
    for (; k < p.splits; ++k) s1 += ry_ld4(p.slabs + (size_t)k * st + i4);
    const f32x4 s = (s0 + s1) + (s2 + s3);
    f32x4 o;
#pragma unroll
    for (int u = 0; u < 4; ++u) o[u] = ry_act(fmaf(s[u], sc[u], sh[u]), p.act, p.slope);
    if (p.out) ry_st4(p.out + i4, o);
    if (p.out16) {
        if (p.x3) ry_st4_bf16_x3(p.out16, (size_t)((unsigned)i4 / (unsigned)p.N), p.N, n, o);
        else ry_st4_bf16(p.out16 + i4, o);
    }
}

// Many slabs, few outputs (the weight-streaming layers at the bottom of the U-Net): 64 float4 columns per workgroup,
// the slabs are divided over 4 waves (wave w sums slabs w, w+4, ...), combined through LDS in a fixed order.
RY_KERNEL(256) void ry_splitk_reduce_wide(RyReduceParams p) {
    __shared__ __attribute__((aligned(16))) float part[4 * 64 * 4];
    const int tid = (int)threadIdx.x, col = tid & 63, grp = tid >> 6;
    const long long i4 = ((long long)blockIdx.x * 64 + col) * 4;
    const bool live = i4 < p.total;
    const size_t st = (size_t)p.slab_stride;
    const int n = (int)((unsigned)i4 % (unsigned)p.N);          // (the executor keeps every activation below 2^31 elements: 32-bit arithmetic)
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (grp == 0 && live) { sc = ry_ld4(p.scale + n); sh = ry_ld4(p.shift + n); }     // requested with the slabs, not behind the barrier
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        // group grp sums slabs grp, grp + 4, ... in that order; eight loads are in flight per round trip (the kernel is a
        // chain of memory latencies: 24..384 workgroups, up to 32 slabs per group)
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        for (int k = grp; k < p.splits; k += 32) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (k + 4 * u < p.splits) ? ry_ld4(p.slabs + (size_t)(k + 4 * u) * st + i4) : zero;
#pragma unroll
            for (int u = 0; u < 8; ++u) a0 += v[u];
        }
    }
    ry_st4(&part[(grp * 64 + col) * 4], a0);
    __syncthreads();
    if (grp == 0 && live) {
        const f32x4 s = (ry_ld4(&part[col * 4]) + ry_ld4(&part[(64 + col) * 4])) + (ry_ld4(&part[(128 + col) * 4]) + ry_ld4(&part[(192 + col) * 4]));
        f32x4 o;
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = ry_act(fmaf(s[u], sc[u], sh[u]), p.act, p.slope);
        if (p.out) ry_st4(p.out + i4, o);
        if (p.out16) {
            if (p.x3) ry_st4_bf16_x3(p.out16, (size_t)((unsigned)i4 / (unsigned)p.N), p.N, n, o);
            else ry_st4_bf16(p.out16 + i4, o);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// ry_c2d_os<MT4, NT4, WAVES, DEPTH, XL> -- stage-2 layer, OUTPUT-STATIONARY form for the weight-streaming bottom of the U-Net (round 5).
// A layer with a handful of output pixels (encoder c6 / c7, decoder c0 / c1 at 300 frames: 12-48 pixels against 16.8-33.5 MB of filters)
// is a stream of filters with a little arithmetic attached.  The implicit GEMM above can only fill the chip with it by cutting K over
// hundreds of workgroups: two chunks each -- all prologue and epilogue -- then raw slabs and a reduce launch of equal length.  Here a
// workgroup owns a tile of 4 MT4 pixels x 4 NT4 output channels over the WHOLE K axis (taps x input channels): no slabs, no reduce
// node, one fixed summation order.  The arithmetic runs on v_mfma_f32_4x4x1_16B_f32 with its sixteen independent blocks mapped to
// sixteen K positions: lane (block b, i) feeds pixel i / output channel i of K position b, so a 16-byte load per lane is 4 K steps for
// a 4-pixel (4-channel) group and 64 input channels per wave-instruction -- a K-batched small-tile GEMM at the full fp32 matrix rate
// whose tile can be as small as 4 x 4 (the 32 x 32 form needs 32 output channels per workgroup, i.e. 16 workgroups for 512 channels).
//   K units of 64 input channels x one tap are dealt to the WAVES waves of the workgroup in contiguous runs; every wave streams its
//   filters ([phase][N / 4][tap][C / 64][lane][4]: one contiguous KiB per (4 channels, unit), consecutive units consecutive) and its
//   activations (NHWC, out-of-image taps read the zero pixel behind the buffer) through a ring of DEPTH units in flight -- straight into
//   registers, or (XL, described at the template) by DMA through a wave-private LDS slot with explicit waits --, no barrier in the K loop;
//   then: reduce-scatter over the sixteen blocks (lane bits 2-5: DPP inside a row of 16, ds_bpermute across), a fixed-order sum over
//   the waves through the LDS, folded BN + activation, dense store.
// The host guarantees (launch_c2d_os): C1, C2 multiples of 256 (a round of four units never straddles a source), N a multiple of 4 NT4,
// units % (4 WAVES) == 0, sources followed by a zeroed pixel.
// ---------------------------------------------------------------------------------------------
struct RyC2dOsParams {
    const float* src1;
    const float* src2;          // second source of a skip concat (channels C1 .. C1 + C2), or null
    const float* wt;            // [phase][N / 4][tap][(C1 + C2) / 64][lane = 4 * ((c % 64) / 4) + n % 4][c % 4]
    const float* scale;
    const float* shift;
    float* out;                 // NHWC, or null when only the bf16 copy is wanted
    unsigned short* out16;      // optional bf16 copy for consumers on the bf16 pipe (null in fp32 mode)
    int x3;                     // out16 format: 0 = [pixel][N] bf16, 1 = split-bf16 [pixel][hi (N) | lo (N)]
    int C1, C2;
    int B, Hi, Wi, Ho, Wo;
    int Mh, Mw, M;              // row grid per image (conv: output pixels, sub-pixel deconv: input pixels), rows in all = B * Mh * Mw
    int stride, pad, ostride;   // as RyConvGeom
    int ntaps, nphases;
    int N;
    int act;
    float slope;
    int mtiles, ntiles;         // tiles of 4 MT4 rows x 4 NT4 channels; 1-D grid, XCD-ordered: the M-tiles of one filter slice run on one XCD
    unsigned zp1, zp2;          // byte offset of the zeroed pixel behind each source
    int kw, dil;                // convolution: taps per kernel row, dilation (tap (ky, kx) reads input offset (ky, kx) * dil); the sub-pixel deconvolution
                                // (ostride == 2) has 4 phases (py, px) of 2 x 2 taps (ty, tx) reading input offset py ? 1 - ty : -ty (likewise x)
    float inv_Mimg, inv_Mw, inv_mtiles, inv_ntiles, inv_cpt, inv_kw;
};

// one step of the reduce-scatter over the K blocks: lanes with bit MASK set keep the upper half of the sums
template <int NV, int MASK>
RY_DEV void ry_rs_step(const float (&v)[NV], float (&h)[NV / 2], int lane) {
    const bool up = (lane & MASK) != 0;
#pragma unroll
    for (int i = 0; i < NV / 2; ++i) {
        float lo = v[i], hi = v[i + NV / 2];
        ry_keep2(lo, hi);
        const float recv = ry_shfl_xor_c<MASK>(up ? lo : hi);
        h[i] = (up ? hi : lo) + recv;
    }
}

// XL = true: the pixels travel global -> LDS by DMA and from there into the MFMA operand registers.  A wave-load of pixels straight into registers has to
// follow the lane order of the instruction -- lane (block b, i) = pixel i, channels 4 b .. 4 b + 3: consecutive lanes read DIFFERENT pixels, 16 bytes each --
// and measured at about half the rate of the filter stream (contiguous KiB).  By DMA, lane 16 i + q of a wave-instruction fetches quad q ^ 4 i of pixel i
// (sixteen consecutive lanes = one contiguous 256-byte piece) and the data lands lane-linear in a wave-private ring slot; lane (b, i) then reads position
// 16 i + (b ^ 4 i) with one ds_read_b128: the XOR on the source side makes the sixteen lanes of every read group hit sixteen different bank quads.
template <int MT4, int NT4, int WAVES, int DEPTH, bool XL>
RY_KERNEL(64 * WAVES) void ry_c2d_os(RyC2dOsParams p) {
    constexpr int MT = 4 * MT4, NT = 4 * NT4;
    constexpr int V = MT4 * NT4 * 4, VP = (V + 15) / 16 * 16, L = VP / 16;      // partial sums per lane, padded for the 16-way scatter
    static_assert(DEPTH * (MT4 + NT4) <= 56, "loads in flight per wave stay below the vmcnt range");
    __shared__ float red[WAVES * VP * 4];

    const int tid = (int)threadIdx.x, lane = tid & 63, wave = ry_uniform(tid >> 6);
    const int blk = lane >> 2, sub = lane & 3;
    const int total = p.nphases * p.ntiles * p.mtiles;
    const int per_xcd = (total + 7) >> 3;
    const int lid = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);      // XCD b % 8 owns a contiguous run of (slice, M-tile) pairs
    if (lid >= total) return;
    const int sl = ry_fdiv(lid, p.mtiles, p.inv_mtiles), mt = lid - sl * p.mtiles;
    const int phase = ry_fdiv(sl, p.ntiles, p.inv_ntiles), nt = sl - phase * p.ntiles;
    const int m0 = mt * MT, n0 = nt * NT;
    const int Mimg = p.Mh * p.Mw;

    constexpr int RND = 4;                     // units per round: a round never straddles a tap or a source (the host checks C1 / 64, C2 / 64, units per wave)
    static_assert(DEPTH == 2 || DEPTH == 4, "two or four units in flight");
    const int cpt1 = p.C1 >> 6, cpt = (p.C1 + p.C2) >> 6;
    const int U = p.ntaps * cpt, nu = U / WAVES, nr = nu / RND;
    const int u0 = wave * nu;
    // filters of (phase, channel group n0 / 4 + h), unit u: one KiB at ((phase * N / 4 + n0 / 4 + h) * U + u) * 256 floats
    const float* const wq0 = p.wt + ((size_t)(phase * (p.N >> 2) + (n0 >> 2)) * (size_t)U + (size_t)u0) * 256 + lane * 4;
    const size_t wh = (size_t)U * 256;

    // Byte offset of the input pixel of (source, tap, row of the tile), or of the source's zero pixel when the tap falls outside the image
    // or the row outside the batch: a table in the LDS, filled once; the K loop reads MT4 entries when its tap or source changes.
    __shared__ unsigned otab[2 * 16 * MT];
    {
        const int per_src = p.ntaps * MT;
        for (int e = tid; e < 2 * per_src; e += 64 * WAVES) {
            const int src = e >= per_src ? 1 : 0, e1 = e - src * per_src;
            const int tap = ry_fdiv(e1, MT, 1.0f / MT), rl = e1 - tap * MT;
            const int r = m0 + rl;
            const int b = ry_fdiv(r, Mimg, p.inv_Mimg), rem = r - b * Mimg;
            const int y = ry_fdiv(rem, p.Mw, p.inv_Mw), x = rem - y * p.Mw;
            int dy, dx;
            if (p.ostride == 2) { const int ty = tap >> 1, tx = tap & 1; dy = (phase >> 1) ? 1 - ty : -ty; dx = (phase & 1) ? 1 - tx : -tx; }
            else { const int ky = ry_fdiv(tap, p.kw, p.inv_kw), kx = tap - ky * p.kw; dy = ky * p.dil; dx = kx * p.dil; }
            const int iy = y * p.stride - p.pad + dy, ix = x * p.stride - p.pad + dx;
            const bool ok = r < p.M && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            const unsigned pix = (unsigned)((b * p.Hi + iy) * p.Wi + ix);
            otab[e] = ok ? pix * (unsigned)((src ? p.C2 : p.C1) * 4) : (src ? p.zp2 : p.zp1);
        }
    }
    ry_lds_barrier();

    // state of the round being REQUESTED (wave-uniform): position in the run, its tap and first chunk, where its filters and pixels start
    // (every wave walks its run from the head: a rotated start -- waves spread over the offsets of their filter regions -- and non-temporal
    // filter loads were measured as nulls, profiles/r05/b_*)
    int r_rel = 0, r_tap = ry_fdiv(u0, cpt, p.inv_cpt), r_chunk = u0 - r_tap * cpt;
    int r_key = -1;
    unsigned pb[MT4];
    const float* xs = p.src1;
    const float* wr = wq0;
    // XL: wave-private ring slots of MT4 KiB each; this lane's DMA role (pixel lane / 16, quad (lane % 16) ^ 4 (lane / 16)) and its read position
    // (DEPTH slots, refilled in place: the reads of a slot have RETURNED -- explicit lgkmcnt(0) in consume -- before the DMA that overwrites it is issued)
    __shared__ __attribute__((aligned(16))) float xs0[XL ? WAVES * MT4 * 256 : 4], xs1[XL ? WAVES * MT4 * 256 : 4];
    __shared__ __attribute__((aligned(16))) float xs2[XL && DEPTH == 4 ? WAVES * MT4 * 256 : 4], xs3[XL && DEPTH == 4 ? WAVES * MT4 * 256 : 4];
    const int dpx = lane >> 4, dq = (lane & 15) ^ (4 * dpx);
    const int rpos = (sub * 16 + (blk ^ (4 * sub))) * 4;            // floats into a group's KiB
    auto round_setup = [&]() {
        const bool first = r_chunk < cpt1;
        const int key = 2 * r_tap + (first ? 0 : 1);
        if (key != r_key) {                                         // another tap or the other source: this lane's pixel offsets
            r_key = key;
            const unsigned* ot = otab + ((first ? 0 : p.ntaps) + r_tap) * MT + (XL ? dpx : sub);
#pragma unroll
            for (int g = 0; g < MT4; ++g) pb[g] = ot[4 * g] + (unsigned)(XL ? dq : blk) * 16u;
        }
        xs = (first ? p.src1 : p.src2 - (size_t)cpt1 * 64) + r_chunk * 64;
        wr = wq0 + (size_t)r_rel * (RND * 256);
    };
    auto round_next = [&]() {
        r_chunk += RND;
        if (r_chunk == cpt) { r_chunk = 0; ++r_tap; }
        ++r_rel;
        round_setup();
    };

    f32x4 xa[XL ? 1 : DEPTH][MT4], wb[DEPTH][NT4];
    f32x4 acc[MT4][NT4];
#pragma unroll
    for (int g = 0; g < MT4; ++g)
#pragma unroll
        for (int h = 0; h < NT4; ++h) acc[g][h] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto issue = [&](int d, int j) {                                // unit j of the round being requested -> slot d
        {
#pragma unroll
            for (int h = 0; h < NT4; ++h) wb[d][h] = ry_ld4(wr + j * 256 + (size_t)h * wh);
        }
        {
            if (XL) {
                float* const slot = (d == 0 ? xs0 : d == 1 ? xs1 : d == 2 ? xs2 : xs3) + wave * (MT4 * 256);
#pragma unroll
                for (int g = 0; g < MT4; ++g) ry_glds16_off(xs + j * 64, pb[g], slot + g * 256);
            } else {
                const char* sb = reinterpret_cast<const char*>(xs + j * 64);
#pragma unroll
                for (int g = 0; g < MT4; ++g) xa[d][g] = *reinterpret_cast<const f32x4*>(sb + pb[g]);
            }
        }
    };
    auto consume = [&](int d, int j, int after) {                   // unit j of the round being multiplied, in slot d; `after` younger units are in flight
        if (XL) {
            const float* const slot = (d == 0 ? xs0 : d == 1 ? xs1 : d == 2 ? xs2 : xs3) + wave * (MT4 * 256) + rpos;
            if (after >= 3) ry_own_dma_landed<3 * (MT4 + NT4)>();
            else if (after == 2) ry_own_dma_landed<2 * (MT4 + NT4)>();
            else if (after == 1) ry_own_dma_landed<MT4 + NT4>();
            else ry_own_dma_landed<0>();
#pragma unroll
            for (int g = 0; g < MT4; ++g) xa[0][g] = ry_ld4(slot + g * 256);
            ry_lds_reads_returned();
        }
        const int xd = XL ? 0 : d;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < MT4; ++g)
#pragma unroll
                for (int h = 0; h < NT4; ++h) acc[g][h] = ry_mfma_4x4x1(xa[xd][g][t], wb[d][h][t], acc[g][h]);
    };

    // Ring of DEPTH units: while unit j of a round is multiplied, unit j + DEPTH is requested -- the last DEPTH requests of a round already
    // belong to the next one, so the round state moves on at j = RND - DEPTH.  The last round requests nothing beyond the run.
    {
        round_setup();
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) issue(d, d);
        for (int i = 1; i < nr; ++i) {
#pragma unroll
            for (int j = 0; j < RND; ++j) {
                if (j == RND - DEPTH) round_next();
                consume(j % DEPTH, j, DEPTH - 1);
                issue(j % DEPTH, (j + DEPTH) % RND);
            }
        }
#pragma unroll
        for (int j = 0; j < RND; ++j) {
            consume(j % DEPTH, j, RND - 1 - j < DEPTH - 1 ? RND - 1 - j : DEPTH - 1);
            if (j + DEPTH < RND) issue(j % DEPTH, j + DEPTH);
        }
    }

    // ---- sum over the sixteen K blocks (reduce-scatter: lane bits 2..5), then over the waves (LDS, fixed order) ----
    float v16[VP];
#pragma unroll
    for (int q = 0; q < VP; ++q) v16[q] = q < V ? acc[(q >> 2) / NT4][(q >> 2) % NT4][q & 3] : 0.f;      // q = (g * NT4 + h) * 4 + r
    float v8[VP / 2], v4[VP / 4], v2[VP / 8], v1[L];
    ry_rs_step<VP, 4>(v16, v8, lane);
    ry_rs_step<VP / 2, 8>(v8, v4, lane);
    ry_rs_step<VP / 4, 16>(v4, v2, lane);
    ry_rs_step<VP / 8, 32>(v2, v1, lane);
    const int q0 = ry_bitrev(blk, 4) * L;                     // this lane now holds the sums q0 .. q0 + L - 1 of column `sub`
#pragma unroll
    for (int u = 0; u < L; ++u) red[(wave * VP + q0 + u) * 4 + sub] = v1[u];
    __syncthreads();
    const int pdy = p.ostride == 2 ? phase >> 1 : 0, pdx = p.ostride == 2 ? phase & 1 : 0;
    for (int idx = tid; idx < MT * NT; idx += 64 * WAVES) {
        const int pxl = idx / NT, c = idx - pxl * NT;
        const int q = ((pxl >> 2) * NT4 + (c >> 2)) * 4 + (pxl & 3), j = c & 3;
        float s = red[q * 4 + j];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) s += red[(w * VP + q) * 4 + j];
        const int r = m0 + pxl;
        if (r < p.M) {
            const int b = ry_fdiv(r, Mimg, p.inv_Mimg), rem = r - b * Mimg;
            const int y = ry_fdiv(rem, p.Mw, p.inv_Mw), x = rem - y * p.Mw;
            const size_t opix = ((size_t)b * p.Ho + y * p.ostride + pdy) * p.Wo + x * p.ostride + pdx;
            const int n = n0 + c;
            const float o = ry_act(fmaf(s, p.scale[n], p.shift[n]), p.act, p.slope);
            if (p.out) p.out[opix * p.N + n] = o;
            if (p.out16) {
                const unsigned short hi = ry_f2bf(o);
                if (p.x3) { p.out16[opix * (size_t)(2 * p.N) + n] = hi; p.out16[opix * (size_t)(2 * p.N) + p.N + n] = ry_f2bf(o - ry_bf2f(hi)); }
                else p.out16[opix * p.N + n] = hi;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Generic direct convolution (VALU): one thread per (output pixel, output channel), channel fastest.
// ---------------------------------------------------------------------------------------------
struct RyDirectParams {
    RyConvGeom g;
    const float* wd;            // [phase][tap][C1+C2][N]
    const float* scale;
    const float* shift;
    float* out;                 // NHWC
    int act;
    float slope;
};

RY_KERNEL(256) void ry_conv_direct(RyDirectParams p) {
    const RyConvGeom& g = p.g;
    const int phase = (int)blockIdx.y;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int Mimg = g.Mh * g.Mw;
    const long long total = (long long)g.B * Mimg * g.N;
    if (idx >= total) return;
    const int n = (int)(idx % g.N);
    const int m = (int)(idx / g.N);
    const int b = m / Mimg, rem = m - b * Mimg;
    const int ry = rem / g.Mw, rx = rem - ry * g.Mw;
    const int yb = ry * g.stride - g.pad, xb = rx * g.stride - g.pad;
    const int Ctot = g.C1 + g.C2;
    float acc = 0.f;
    for (int t = 0; t < g.ntaps; ++t) {
        const int iy = yb + g.tdy[phase][t], ix = xb + g.tdx[phase][t];
        if ((unsigned)iy >= (unsigned)g.Hi || (unsigned)ix >= (unsigned)g.Wi) continue;
        const size_t pix = (size_t)(b * g.Hs + iy) * g.Wi + ix;
        const float* w = p.wd + ((size_t)(phase * g.ntaps + t) * Ctot) * g.N + n;
        const float* s1 = g.src1 + pix * g.C1;
        for (int c = 0; c < g.C1; ++c) acc = fmaf(s1[c], w[(size_t)c * g.N], acc);
        if (g.C2 > 0) {
            const float* s2 = g.src2 + pix * g.C2;
            const float* w2 = w + (size_t)g.C1 * g.N;
            for (int c = 0; c < g.C2; ++c) acc = fmaf(s2[c], w2[(size_t)c * g.N], acc);
        }
    }
    const size_t ob = (size_t)(b * g.Hos + ry * g.ostride + g.pdy[phase]) * g.Wo + rx * g.ostride + g.pdx[phase];
    p.out[ob * g.N + n] = ry_act(fmaf(acc, p.scale[n], p.shift[n]), p.act, p.slope);
}

// ---------------------------------------------------------------------------------------------
// Stage-2 end layers.  encoder c0: 1 -> N channels, 3x3 (HBM-write bound: one 16-byte store per lane).
// decoder c7: (C1 + C2) -> 1 channel, 3x3 over the un-materialised concat of two sources, with the
// SuperResolution.convert post-processing (exp, edge-pad of the dropped bin, crop) fused in.
// ---------------------------------------------------------------------------------------------
struct RySrFirstParams {
    const float* x;             // [B][H][W]
    const float* w;             // [9][N]
    const float* scale;
    const float* shift;
    float* out;                 // [B][H][W][N]
    unsigned short* out16;      // optional bf16 copy (consumers on the bf16 path)
    int x3;                     // out16 format: 0 = plain bf16, 1 = split-bf16 [pixel][hi | lo] (ry_st4_bf16_x3)
    int B, H, W, N;
    int act;
    float slope;
    int qshift;                 // log2(N / 4) when that is a power of two, else -1
};

template <int PX>
RY_KERNEL(256) void ry_sr_first(RySrFirstParams p) {     // grid: (pixel groups x channel quads of one row / 256, rows, windows)
    const int quads = p.N >> 2;
    const int gpr = (p.W + PX - 1) / PX;                       // pixel groups per row
    const int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
    // [r2] the row and the window come from the grid and the channel quad from a shift when N / 4 is a power of two: the kernel was
    // VALU-bound on four emulated integer divisions per thread (16.4 us for a 50 MB write)
    const int gx = p.qshift >= 0 ? idx >> p.qshift : idx / quads;
    const int cq = idx - gx * quads;
    if (gx >= gpr) return;
    const int y = (int)blockIdx.y, b = (int)blockIdx.z;
    const int x0 = gx * PX;
    f32x4 wr[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[t] = ry_ld4(p.w + (size_t)t * p.N + cq * 4);
    const f32x4 sc = ry_ld4(p.scale + cq * 4), sh = ry_ld4(p.shift + cq * 4);
    float xv[3][PX + 2];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int iy = y + dy - 1;
#pragma unroll
        for (int j = 0; j < PX + 2; ++j) {
            const int ix = x0 + j - 1;
            xv[dy][j] = ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? p.x[((size_t)b * p.H + iy) * p.W + ix] : 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        if (x0 + j >= p.W) break;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float xs_ = xv[t / 3][j + t % 3];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = fmaf(xs_, wr[t][u], acc[u]);
        }
        f32x4 o;
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = ry_act(fmaf(acc[u], sc[u], sh[u]), p.act, p.slope);
        const size_t oi = (((size_t)b * p.H + y) * p.W + x0 + j) * p.N + cq * 4;
        if (p.out) ry_st4(p.out + oi, o);
        if (p.out16) {
            if (p.x3) ry_st4_bf16_x3(p.out16, ((size_t)b * p.H + y) * p.W + x0 + j, p.N, cq * 4, o);
            else ry_st4_bf16(p.out16 + oi, o);
        }
    }
}

struct RySrLastParams {
    const float* src1;
    const float* src2;
    int C1, C2;
    const float* w;             // [9][C1+C2]
    const float* scale;         // [1]
    const float* shift;         // [1]
    float* out;                 // [B][out_rows][out_cols]
    int B, H, W;
    int rows_valid;             // number of output rows computed: rows row0 .. row0 + rows_valid - 1 (the padding behind the real frames is never computed)
    int row0;                   // first output row (> 0 when the caller discards the leading frames of the window, ry_sr_convert_rows)
    int out_rows;               // rows per image of `out` (the real frames of the window)
    int out_cols;               // W, or W + 1 with the last bin repeated (pad mode 'edge')
    int do_exp;
    int x3;                     // sources are split-bf16 copies [pixel][hi | lo] (rolling form only)
    int xcd_band;               // rolling form: 1 = each XCD takes a contiguous band of output rows (the grid is then padded to 8 x ceil(blocks / 8))
};

// Simple form: 32 lanes per output pixel, 9 taps gathered per pixel (any width).
RY_KERNEL(256) void ry_sr_last_gather(RySrLastParams p) {
    const int l = (int)threadIdx.x & 31;
    const long long pix = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);
    const long long total = (long long)p.B * p.rows_valid * p.W;
    const bool live = pix < total;
    const long long pp = live ? pix : 0;
    const int x = (int)(pp % p.W);
    const int y = (int)((pp / p.W) % p.rows_valid) + p.row0;
    const int b = (int)(pp / ((long long)p.W * p.rows_valid));
    const int Ctot = p.C1 + p.C2;
    float acc = 0.f;
    if (live) {
        for (int t = 0; t < 9; ++t) {
            const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
            if ((unsigned)iy >= (unsigned)p.H || (unsigned)ix >= (unsigned)p.W) continue;
            const size_t ipx = ((size_t)b * p.H + iy) * p.W + ix;
            for (int c = l * 4; c < Ctot; c += 128) {
                const f32x4 v = (c < p.C1) ? ry_ld4(p.src1 + ipx * p.C1 + c) : ry_ld4(p.src2 + ipx * p.C2 + (c - p.C1));
                const f32x4 wv = ry_ld4(p.w + (size_t)t * Ctot + c);
                acc = fmaf(v[3], wv[3], fmaf(v[2], wv[2], fmaf(v[1], wv[1], fmaf(v[0], wv[0], acc))));
            }
        }
    }
    acc += ry_shfl_xor(acc, 16);
    acc += ry_shfl_xor(acc, 8);
    acc += ry_shfl_xor(acc, 4);
    acc += ry_shfl_xor(acc, 2);
    acc += ry_shfl_xor(acc, 1);
    if (live && l == 0) {
        float v = fmaf(acc, p.scale[0], p.shift[0]);
        if (p.do_exp) v = expf(v);
        float* o = p.out + ((size_t)b * p.out_rows + y) * p.out_cols;
        o[x] = v;
        if (x == p.W - 1 && p.out_cols > p.W) o[p.W] = v;
    }
}

// Rolling form (C1 + C2 == 128, W % 16 == 0): a 32-lane group owns a strip of 16 output pixels of one row;
// lane = 4 channels.  Every input pixel vector of the 3 x 18 halo is loaded ONCE, multiplied by the three
// kx taps of its row and accumulated into the outputs it touches; the 16 per-lane partial sums are then
// reduced across the 32 lanes with a transposing (reduce-scatter) butterfly: 15 + 1 shuffles instead of 80.
// X3 = true (split-bf16 mode): the two sources are the producers' split-bf16 copies [pixel][hi (Cs) | lo (Cs)] (ry_st4_bf16_x3) and a
// lane rebuilds its four channels as hi + lo -- the same bytes per pixel as fp32, and neither producer has to write an fp32
// copy for this layer alone.
template <bool X3>
RY_KERNEL(256, 2) void ry_sr_last(RySrLastParams p) {
    constexpr int SW = 16;
    const int l = (int)threadIdx.x & 31;
    const int strips = p.W / SW;
    const long long total = (long long)p.B * p.rows_valid * strips;
    // Workgroup b runs on XCD b % 8 (one L2 each) and every input row is read by three output rows: give each XCD a contiguous
    // band of output rows, so that the re-reads hit ITS L2 instead of going back to the fabric three times (measured FETCH_SIZE
    // 109 MB raw per launch against 100 MB of input with raster-order blocks, i.e. ~2.2x in HBM bytes).
    long long lb = (long long)blockIdx.x;
    if (p.xcd_band) {
        const long long nb = (total + 7) >> 3, per = (nb + 7) >> 3;
        lb = (long long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if (lb >= nb) return;
    }
    const long long sid = lb * 8 + (threadIdx.x >> 5);
    const bool live = sid < total;
    const long long ss = live ? sid : 0;
    const int x0 = (int)(ss % strips) * SW;
    const int y = (int)((ss / strips) % p.rows_valid) + p.row0;
    const int b = (int)(ss / ((long long)strips * p.rows_valid));
    const int c = l * 4;
    const bool first = c < p.C1;
    const int Cs = first ? p.C1 : p.C2;
    const float* src = first ? p.src1 : p.src2;
    const int cl = first ? c : c - p.C1;            // this lane's first channel inside its source
    // four channels of pixel `px` of a source row: fp32 -> one 16-byte load; split-bf16 -> hi and lo halves, 8 bytes each
    auto ldpx = [&](const float* row, int px) -> f32x4 {
        if (X3) {
            const unsigned short* q = reinterpret_cast<const unsigned short*>(row) + (size_t)px * (size_t)(2 * Cs) + cl;
            const u16x4 h = *reinterpret_cast<const u16x4*>(q), lo = *reinterpret_cast<const u16x4*>(q + Cs);
            f32x4 v;
            v[0] = ry_bf2f(h[0]) + ry_bf2f(lo[0]); v[1] = ry_bf2f(h[1]) + ry_bf2f(lo[1]);
            v[2] = ry_bf2f(h[2]) + ry_bf2f(lo[2]); v[3] = ry_bf2f(h[3]) + ry_bf2f(lo[3]);
            return v;
        }
        return ry_ld4(row + (size_t)px * Cs + cl);
    };
    float acc[SW + 2];                             // acc[j] = output column x0 - 1 + j (two halo slots are discarded)
#pragma unroll
    for (int j = 0; j < SW + 2; ++j) acc[j] = 0.f;
    if (live) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = y + ky - 1;
            const bool rok = (unsigned)iy < (unsigned)p.H;
            const int iyc = rok ? iy : y;                       // padded rows read a valid row and are zeroed by `rz`
            const float rz = rok ? 1.f : 0.f;
            f32x4 w0 = ry_ld4(p.w + (size_t)(ky * 3 + 0) * 128 + c);
            f32x4 w1 = ry_ld4(p.w + (size_t)(ky * 3 + 1) * 128 + c);
            f32x4 w2 = ry_ld4(p.w + (size_t)(ky * 3 + 2) * 128 + c);
            w0 *= rz; w1 *= rz; w2 *= rz;
            const float* row = src + ((size_t)b * p.H + iyc) * p.W * Cs;      // a split-bf16 pixel is 2 Cs bf16 = Cs floats too
            // interior columns x0 .. x0+SW-1 are always inside the image (W % SW == 0): no tests, loads batch freely
#pragma unroll
            for (int j = 1; j <= SW; ++j) {
                const f32x4 v = ldpx(row, x0 - 1 + j);
                // out[ox] += w[kx] . in[ox + kx - 1]  ->  input ix feeds ox = ix + 1 - kx, i.e. acc[j + 1 - kx]
                acc[j + 1] += fmaf(v[3], w0[3], fmaf(v[2], w0[2], fmaf(v[1], w0[1], v[0] * w0[0])));
                acc[j]     += fmaf(v[3], w1[3], fmaf(v[2], w1[2], fmaf(v[1], w1[1], v[0] * w1[0])));
                acc[j - 1] += fmaf(v[3], w2[3], fmaf(v[2], w2[2], fmaf(v[1], w2[1], v[0] * w2[0])));
            }
            // the two halo columns x0-1 and x0+SW (zero outside the image)
            {
                const bool lok = x0 > 0, rgt = x0 + SW < p.W;
                const f32x4 vl = ldpx(row, lok ? x0 - 1 : x0);
                const f32x4 vr = ldpx(row, rgt ? x0 + SW : x0);
                const float zl = lok ? 1.f : 0.f, zr = rgt ? 1.f : 0.f;
                acc[1]  += zl * fmaf(vl[3], w0[3], fmaf(vl[2], w0[2], fmaf(vl[1], w0[1], vl[0] * w0[0])));   // ix = x0-1 feeds ox = x0 via kx = 0
                acc[SW] += zr * fmaf(vr[3], w2[3], fmaf(vr[2], w2[2], fmaf(vr[1], w2[1], vr[0] * w2[0])));   // ix = x0+SW feeds ox = x0+SW-1 via kx = 2
            }
        }
    }
    // reduce-scatter over the 32 lanes: after the steps with masks 16, 8, 4, 2 a lane holds ONE output column
    float r8[8], r4[4], r2[2], r1;
    const bool h16 = (l & 16) != 0, h8 = (l & 8) != 0, h4 = (l & 4) != 0, h2 = (l & 2) != 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float lo = acc[1 + i], hi = acc[1 + i + 8];
        const float recv = ry_shfl_xor(h16 ? lo : hi, 16);
        r8[i] = (h16 ? hi : lo) + recv;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float recv = ry_shfl_xor(h8 ? r8[i] : r8[i + 4], 8);
        r4[i] = (h8 ? r8[i + 4] : r8[i]) + recv;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float recv = ry_shfl_xor(h4 ? r4[i] : r4[i + 2], 4);
        r2[i] = (h4 ? r4[i + 2] : r4[i]) + recv;
    }
    {
        const float recv = ry_shfl_xor(h2 ? r2[0] : r2[1], 2);
        r1 = (h2 ? r2[1] : r2[0]) + recv;
    }
    r1 += ry_shfl_xor(r1, 1);
    if (live && (l & 1) == 0) {
        const int col = (h16 ? 8 : 0) + (h8 ? 4 : 0) + (h4 ? 2 : 0) + (h2 ? 1 : 0);
        const int x = x0 + col;
        float v = fmaf(r1, p.scale[0], p.shift[0]);
        if (p.do_exp) v = expf(v);
        float* o = p.out + ((size_t)b * p.out_rows + y) * p.out_cols;
        o[x] = v;
        if (x == p.W - 1 && p.out_cols > p.W) o[p.W] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Stage-1: weight-streaming 1-D conv / deconv with deferred producer epilogue.
// ---------------------------------------------------------------------------------------------
struct RySrc1d {
    const float* raw;           // [splits][B*L][Craw] raw conv sums (or a plain activation, splits=1)
    const float* scale;         // [Craw] or null (identity)
    const float* shift;         // [Craw] or null
    long long slab_stride;
    int C;                      // channels this source contributes (GLU: Craw = 2*C)
    int Craw;
    int splits;
    int act;
};

RY_DEV float ry_src1d_load(const RySrc1d& s, size_t pix, int c, float slope) {
    const float* q = s.raw + pix * s.Craw + c;
    float v = q[0];
    for (int k = 1; k < s.splits; ++k) v += q[(size_t)k * (size_t)s.slab_stride];
    if (s.scale) v = fmaf(v, s.scale[c], s.shift[c]);
    if (s.act == RY_ACT_GLU) {
        const float* qg = q + s.C;
        float gte = qg[0];
        for (int k = 1; k < s.splits; ++k) gte += qg[(size_t)k * (size_t)s.slab_stride];
        if (s.scale) gte = fmaf(gte, s.scale[c + s.C], s.shift[c + s.C]);
        return v * ry_sigmoid(gte);
    }
    return ry_act(v, s.act, slope);
}

// Four consecutive positions of one channel at once: the split-K partial sums of the producer are
// four independent load streams (same split order as ry_src1d_load, so results are identical).
RY_DEV f32x4 ry_src1d_load4(const RySrc1d& s, long long pix0, int c, int valid_mask, float slope) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (valid_mask == 0) return v;
    if (s.act == RY_ACT_GLU) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (valid_mask & (1 << u)) v[u] = ry_src1d_load(s, (size_t)(pix0 + u), c, slope);
        return v;
    }
    // masked-off positions (zero padding, tile tail) alias a valid stream and are zeroed afterwards
    int first = 0;
    while (!(valid_mask & (1 << first))) ++first;
    const float* q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int uu = (valid_mask & (1 << u)) ? u : first;
        q[u] = s.raw + (size_t)(pix0 + uu) * (size_t)s.Craw + c;
    }
    float a0 = q[0][0], a1 = q[1][0], a2 = q[2][0], a3 = q[3][0];
    int k = 1;
    for (; k + 7 < s.splits; k += 8) {             // 32 independent loads in flight; additions stay in split order
        float b0[8], b1[8], b2[8], b3[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t o = (size_t)(k + u) * (size_t)s.slab_stride;
            b0[u] = q[0][o]; b1[u] = q[1][o]; b2[u] = q[2][o]; b3[u] = q[3][o];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { a0 += b0[u]; a1 += b1[u]; a2 += b2[u]; a3 += b3[u]; }
    }
    for (; k + 3 < s.splits; k += 4) {
        const size_t o0 = (size_t)k * (size_t)s.slab_stride, o1 = o0 + (size_t)s.slab_stride;
        const size_t o2 = o1 + (size_t)s.slab_stride, o3 = o2 + (size_t)s.slab_stride;
        const float b00 = q[0][o0], b10 = q[1][o0], b20 = q[2][o0], b30 = q[3][o0];
        const float b01 = q[0][o1], b11 = q[1][o1], b21 = q[2][o1], b31 = q[3][o1];
        const float b02 = q[0][o2], b12 = q[1][o2], b22 = q[2][o2], b32 = q[3][o2];
        const float b03 = q[0][o3], b13 = q[1][o3], b23 = q[2][o3], b33 = q[3][o3];
        a0 = ((a0 + b00) + b01) + b02 + b03; a1 = ((a1 + b10) + b11) + b12 + b13;
        a2 = ((a2 + b20) + b21) + b22 + b23; a3 = ((a3 + b30) + b31) + b32 + b33;
    }
    for (; k < s.splits; ++k) {
        const size_t o = (size_t)k * (size_t)s.slab_stride;
        a0 += q[0][o]; a1 += q[1][o]; a2 += q[2][o]; a3 += q[3][o];
    }
    float sc = 1.f, sh = 0.f;
    if (s.scale) { sc = s.scale[c]; sh = s.shift[c]; }
    v[0] = (valid_mask & 1) ? ry_act(fmaf(a0, sc, sh), s.act, slope) : 0.f;
    v[1] = (valid_mask & 2) ? ry_act(fmaf(a1, sc, sh), s.act, slope) : 0.f;
    v[2] = (valid_mask & 4) ? ry_act(fmaf(a2, sc, sh), s.act, slope) : 0.f;
    v[3] = (valid_mask & 8) ? ry_act(fmaf(a3, sc, sh), s.act, slope) : 0.f;
    return v;
}


struct RyConv1dParams {
    RySrc1d s[2];
    int B, Lin, Lout;
    int Ctot;                   // s[0].C + s[1].C
    int N;                      // output channels
    const float* wd;            // [Ctot][N][4]  (taps beyond k are zero)
    int stride, pad, dil;       // conv modes (deconv is k4 s2 p1)
    float* out;                 // raw slabs [splits][B*Lout][N]
    int splits;
    long long slab_stride;
    float slope;
};

// positions of the input tile a wave needs for TL (conv) / TQ (deconv) outputs
template <int MODE> struct RyC1dTile;
template <> struct RyC1dTile<RY_C1D_S2> { static constexpr int TL = 16, PP = 36; };
template <> struct RyC1dTile<RY_C1D_S1> { static constexpr int TL = 16, PP = 20; };
template <> struct RyC1dTile<RY_C1D_DECONV> { static constexpr int TL = 8, PP = 12; };
template <> struct RyC1dTile<RY_C1D_GEN> { static constexpr int TL = 16, PP = 132; };

// Workgroup = up to 4 waves; wave w owns output channels [(blockIdx.x*4 + w)*64, +64) and all waves share
// the LDS-staged input tile (same positions, same input-channel split).
template <int MODE>
RY_KERNEL(256) void ry_conv1d_ws(RyConv1dParams p) {
    constexpr int TL = RyC1dTile<MODE>::TL, PP = RyC1dTile<MODE>::PP, CS = 32;
    constexpr int NACC = (MODE == RY_C1D_DECONV) ? 2 * TL : TL;
    __shared__ __attribute__((aligned(16))) float xs[CS * PP];

    const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;
    const int lane = tid & 63;
    const int co = ((int)blockIdx.x * 4 + (tid >> 6)) * 64 + lane;
    const int Lr = (MODE == RY_C1D_DECONV) ? p.Lin : p.Lout;         // row axis the tiles walk
    const int tiles = (Lr + TL - 1) / TL;
    const int b = (int)blockIdx.y / tiles;
    const int l0 = ((int)blockIdx.y % tiles) * TL;
    const int split = (int)blockIdx.z;
    const int ci_begin = (int)(((long long)p.Ctot * split) / p.splits);
    const int ci_end = (int)(((long long)p.Ctot * (split + 1)) / p.splits);
    // first input position of the tile
    int pos0, npos;
    if (MODE == RY_C1D_DECONV) { pos0 = l0 - 1; npos = TL + 2; }
    else if (MODE == RY_C1D_S2) { pos0 = l0 * 2 - 1; npos = 2 * TL + 2; }
    else if (MODE == RY_C1D_S1) { pos0 = l0 - p.pad; npos = TL + 3; }
    else { pos0 = l0 * p.stride - p.pad; npos = (TL - 1) * p.stride + 3 * p.dil + 1; }

    float acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = 0.f;
    const bool co_ok = co < p.N;
    const int cw = co_ok ? co : 0;

    // Filters of 4 input channels (4 x 16 bytes per lane) are in flight at a time.  16 were no faster when the predictor runs
    // alone (it is bound by launch and staging latencies) but cost 264 VGPRs per wave: a stage-1 workgroup then takes half of
    // every SIMD's register file and the second stage-2 workgroup of that CU has to wait for it.  At 125 VGPRs the two
    // predictors co-reside and a step is 1.34 instead of 1.36 ms.
    constexpr int WB = 4;
    for (int cc = ci_begin; cc < ci_end; cc += CS) {
        const int cn = (ci_end - cc < CS) ? (ci_end - cc) : CS;
        // the first filter batch does not depend on the staged tile: request it before the staging loads so both
        // latencies overlap
        f32x4 wnext[WB];
#pragma unroll
        for (int u = 0; u < WB; ++u) {
            const int cl = u < cn ? u : cn - 1;
            wnext[u] = ry_ld4(p.wd + ((size_t)(cc + cl) * p.N + cw) * 4);
        }
        __syncthreads();
        // stage xs[cl][pos] = act(scale * sum_splits(raw) + shift) of the producer layer(s), 0 outside [0, Lin)
        for (int e = tid; e < CS * (PP / 4); e += nthr) {
            const int cl = e % CS, p4 = e / CS;
            const int ci = cc + cl;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ci < ci_end) {
                int mask = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pl = p4 * 4 + u;
                    const int pos = pos0 + pl;
                    if (pl < npos && pos >= 0 && pos < p.Lin) mask |= 1 << u;
                }
                // pix0 may be "negative" for masked-off leading positions; only masked-in streams are dereferenced
                const long long pix0 = (long long)b * p.Lin + pos0 + p4 * 4;
                v = (ci < p.s[0].C) ? ry_src1d_load4(p.s[0], pix0, ci, mask, p.slope)
                                    : ry_src1d_load4(p.s[1], pix0, ci - p.s[0].C, mask, p.slope);
            }
            ry_st4(&xs[cl * PP + p4 * 4], v);
        }
        __syncthreads();
        for (int cb = 0; cb < cn; cb += WB) {
            f32x4 wcur[WB];
#pragma unroll
            for (int u = 0; u < WB; ++u) wcur[u] = wnext[u];
            if (cb + WB < cn) {
#pragma unroll
                for (int u = 0; u < WB; ++u) {
                    const int cl = cb + WB + u < cn ? cb + WB + u : cn - 1;
                    wnext[u] = ry_ld4(p.wd + ((size_t)(cc + cl) * p.N + cw) * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < WB; ++u) {
                const int cl = cb + u;
                if (cl >= cn) break;
                const f32x4 w = wcur[u];
                const float* x = &xs[cl * PP];
                if (MODE == RY_C1D_S2) {
                    float xr[PP];
#pragma unroll
                    for (int q = 0; q < PP / 4; ++q) { const f32x4 t = ry_ld4(x + 4 * q); xr[4*q] = t[0]; xr[4*q+1] = t[1]; xr[4*q+2] = t[2]; xr[4*q+3] = t[3]; }
#pragma unroll
                    for (int j = 0; j < TL; ++j)
                        acc[j] = fmaf(w[3], xr[2*j+3], fmaf(w[2], xr[2*j+2], fmaf(w[1], xr[2*j+1], fmaf(w[0], xr[2*j], acc[j]))));
                } else if (MODE == RY_C1D_S1) {
                    float xr[PP];
#pragma unroll
                    for (int q = 0; q < PP / 4; ++q) { const f32x4 t = ry_ld4(x + 4 * q); xr[4*q] = t[0]; xr[4*q+1] = t[1]; xr[4*q+2] = t[2]; xr[4*q+3] = t[3]; }
#pragma unroll
                    for (int j = 0; j < TL; ++j)
                        acc[j] = fmaf(w[3], xr[j+3], fmaf(w[2], xr[j+2], fmaf(w[1], xr[j+1], fmaf(w[0], xr[j], acc[j]))));
                } else if (MODE == RY_C1D_DECONV) {
                    float xr[PP];
#pragma unroll
                    for (int q = 0; q < PP / 4; ++q) { const f32x4 t = ry_ld4(x + 4 * q); xr[4*q] = t[0]; xr[4*q+1] = t[1]; xr[4*q+2] = t[2]; xr[4*q+3] = t[3]; }
                    // out[2q]   = w1*x[q] + w3*x[q-1] ;  out[2q+1] = w0*x[q+1] + w2*x[q]   (xr[j] = x[l0-1+j])
#pragma unroll
                    for (int j = 0; j < TL; ++j) {
                        acc[2*j]   = fmaf(w[3], xr[j],   fmaf(w[1], xr[j+1], acc[2*j]));
                        acc[2*j+1] = fmaf(w[2], xr[j+1], fmaf(w[0], xr[j+2], acc[2*j+1]));
                    }
                } else {
                    for (int j = 0; j < TL; ++j) {
                        float a = acc[j];
#pragma unroll
                        for (int k = 0; k < 4; ++k) a = fmaf(w[k], x[j * p.stride + k * p.dil], a);
                        acc[j] = a;
                    }
                }
            }
        }
    }
    if (!co_ok) return;
    float* outp = p.out + (size_t)split * (size_t)p.slab_stride;
    if (MODE == RY_C1D_DECONV) {
#pragma unroll
        for (int j = 0; j < 2 * TL; ++j) {
            const int l = 2 * l0 + j;
            if (l < p.Lout) outp[((size_t)b * p.Lout + l) * p.N + co] = acc[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < TL; ++j) {
            const int l = l0 + j;
            if (l < p.Lout) outp[((size_t)b * p.Lout + l) * p.N + co] = acc[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// ry_c1d_os<MODE, CB, TP> -- stage-1 layer, OUTPUT-STATIONARY form (the predictor's default since round 2).
// The weight-streaming kernel above splits the input channels over workgroups and leaves up to 32 raw slabs per layer for the
// consumer to re-sum: 17.8 MB of slab traffic per forward against 0.03 MB of activations, and a chain of dependent slab loads
// in front of every layer's first FMA (11-15 us per layer at a 3.4 us launch floor).  Here the LANES run over the input
// channels instead: a workgroup owns CB output channels x (PG x TP) positions, thread (ci lane, position group) loads its
// input column x[pos][ci] (4-byte loads, coalesced over the lanes) and the CB filter taps of its channel (16 bytes each,
// filters stored [co][ci][4]) -- every load of the layer is independent and issued up front: ONE memory round trip -- then
// CB x TP x 4 FMAs per channel, a reduce-scatter butterfly over the 64 lanes (A - 1 + log2(64 / A) shuffles for A = CB x TP
// sums), a fixed-order sum over the waves through the LDS, folded BN + activation, and a dense activated store.  No slabs, no
// deferred epilogue, no second pass; the last layer crops to the real frames as it stores (no ry_materialize node).
//   MODE S2: k4 s2 p1 conv; S1: stride-1 conv, k <= 4 (taps beyond k are zero in the filter), any pad <= 3; DECONV: k4 s2 p1
//   transposed conv in sub-pixel form (TP input positions -> 2 TP outputs).
//   kt_waves = waves that share one position group (1 / 2 / 4 for <= 64 / <= 128 / more input channels); PG = 4 / kt_waves.
// ---------------------------------------------------------------------------------------------
struct RyC1dOsParams {
    const float* sa;            // [B][Lin][Ca] dense, activated
    const float* sb;            // [B][Lin][Cb] second source of a skip concat, or null (Cb = 0)
    int Ca, Cb;
    const float* w;             // [N][Ca + Cb][4]
    const float* scale;         // [N] folded BN scale
    const float* shift;         // [N]
    float* out;                 // [B][keep][N]
    int B, Lin, Lout, N;
    int keep;                   // rows stored per window (Lout, or the real frames of the convert wrapper for the last layer)
    int pad;                    // MODE S1 only
    int act;
    float slope;
    int tiles;                  // position tiles per window (one tile = PG x TP rows) = gridDim.y; gridDim.z = windows
    int kt_shift;               // log2 of the waves that share one position group (0 / 1 / 2 for <= 64 / <= 128 / more input channels)
    int n_real;                 // PADMIN instantiations: real rows per window in `sa`; rows n_real .. Lin - 1 are the per-channel minimum
    unsigned long long* dbg;    // diagnostics (-DRY_S1_STAMPS builds): per-workgroup s_memtime stamps at the phase boundaries, else null
};


// Reduce-scatter over the 64 lanes of a wave: every lane brings N partial sums, the step with mask m halves them (lanes with bit m
// set keep the upper half), masks 1, 2, 4, 8, 16, 32 in that order -- the steps that move the most values are the in-row DPP ones.
// Lane L returns the 64-lane total of sum number bitrev_{log2 N}(L & (N - 1)) (all lanes that share those low bits hold the same
// total).  N - 1 + (6 - log2 N) exchanges, fixed order.
template <int N, int MASK>
struct RyReduceScatter64 {
    RY_DEV_STATIC float run(const float (&v)[N], int lane) {
        float h[N / 2];
        const bool up = (lane & MASK) != 0;
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            float lo = v[i], hi = v[i + N / 2];
            ry_keep2(lo, hi);
            const float recv = ry_shfl_xor_c<MASK>(up ? lo : hi);
            h[i] = (up ? hi : lo) + recv;
        }
        return RyReduceScatter64<N / 2, MASK * 2>::run(h, lane);
    }
};
template <int MASK>
struct RyReduceScatter64<1, MASK> {
    RY_DEV_STATIC float run(const float (&v)[1], int lane) {
        float r = v[0];
        if constexpr (MASK <= 32) {
            r += ry_shfl_xor_c<MASK>(r);
            const float one[1] = {r};
            return RyReduceScatter64<1, MASK * 2>::run(one, lane);
        }
        return r;
    }
};

// USRC: every wave reads ONE source (no second source, or the first one ends on a multiple of 64 channels -- the U-Net's case): the
// source, its row pitch and the row offsets are then wave-uniform and live in SGPRs, and a load costs one VALU add instead of a
// compare / select / 32-bit multiply / 64-bit add chain per lane.  (Measured with s_memtime stamps, RY_S1_STAMPS: the stretch from
// kernel start to the last load issued was ~2000-2500 shader cycles per layer whether or not the instruction cache was warm, and
// ~1000 more per extra channel set: three waves per SIMD x 40 loads x ~6 VALU instructions, a quarter-rate v_mul_lo_u32 among them.)
template <int MODE, int CB, int TP, bool PADMIN, bool USRC>
RY_KERNEL(256) void ry_c1d_os(RyC1dOsParams p) {
    constexpr int TPO = MODE == RY_C1D_DECONV ? 2 * TP : TP;                       // outputs per position group
    constexpr int NP = MODE == RY_C1D_S2 ? 2 * TP + 2 : MODE == RY_C1D_DECONV ? TP + 2 : TP + 3;
    constexpr int A = CB * TPO;
    constexpr int LOG2A = A == 32 ? 5 : A == 16 ? 4 : A == 8 ? 3 : A == 4 ? 2 : A == 2 ? 1 : 0;
    static_assert(A <= 32 && (A & (A - 1)) == 0, "at most 32 sums per thread, a power of two");
    static_assert(!PADMIN || MODE == RY_C1D_S1, "the fused pad is a first-layer (stride-1) feature");
    __shared__ float red[4 * 32];

    const int tid = (int)threadIdx.x, lane = tid & 63, wave = ry_uniform(tid >> 6);
#if defined(RY_HOST_EMU) || !defined(RY_S1_STAMPS)
#define RY_OS_STAMP(i)
#else       // diagnostic build (-DRY_S1_STAMPS): s_memtime at the phase boundaries, one record per workgroup
    unsigned long long* const dbgp = p.dbg ? p.dbg + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 : nullptr;
#define RY_OS_STAMP(i) if (dbgp && tid == 0) dbgp[i] = __builtin_amdgcn_s_memtime();
#endif
    RY_OS_STAMP(0)
    const int ktw = 1 << p.kt_shift, PG = 4 >> p.kt_shift;                           // no integer division in the prologue
    const int pg = wave >> p.kt_shift, kw = wave & (ktw - 1);
    const int Ctot = p.Ca + p.Cb;
    const int co0 = (int)blockIdx.x * CB;
    const int b = (int)blockIdx.z, tile = (int)blockIdx.y;
    const int r0 = (tile * PG + pg) * TP;                                            // first row of this position group
    const int pos0 = MODE == RY_C1D_S2 ? 2 * r0 - 1 : MODE == RY_C1D_DECONV ? r0 - 1 : r0 - p.pad;
    // PADMIN (first layer of the convert wrapper): the source holds n_real rows per window and rows n_real .. Lin - 1 are
    // numpy.pad(mode='minimum'): the per-channel minimum over the real rows, taken here by the workgroups that reach that far
    const int src_rows = PADMIN ? p.n_real : p.Lin;
    const bool need_min = PADMIN && pos0 + NP > p.n_real;                            // wave-uniform
    // [r5] The minimum is taken by the WHOLE workgroup at once: thread (channel c = tid % 64, row group tid / 64 of four) walks rows rg, rg + 4, ...
    // of its channel with sixteen loads in flight, the four partial minima meet in the LDS.  (Round 2 had every lane of a wave walk all
    // real rows of its channel eight at a time -- 38 dependent rounds at 300 frames, with 9 of 64 lanes at work: 16.4 us for the layer, a fifth of
    // the stage-1 forward.)  The condition is workgroup-uniform: the last position group of the tile reaches the padding.
    __shared__ float cmins[4 * 64];
    if (PADMIN) {
        const int r_last = (tile * PG + PG - 1) * TP - p.pad + NP;                   // one past the last input row any wave of this workgroup reads
        if (r_last > p.n_real) {
            const int c = lane, rg = wave;                                           // (the fused pad needs Ca <= 64, one lane set: build_plan)
            const float* colc = p.sa + (size_t)b * (size_t)p.n_real * (size_t)p.Ca + (c < p.Ca ? c : 0);
            float m = INFINITY;
            for (int r = rg; r < p.n_real; r += 64) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = colc[(unsigned)((r + 4 * u < p.n_real ? r + 4 * u : rg) * p.Ca)];
#pragma unroll
                for (int u = 0; u < 16; ++u) m = fminf(m, v[u]);
            }
            cmins[rg * 64 + c] = m;
            __syncthreads();
        }
    }

    // the epilogue's scale / shift of the output this thread will store: requested now, with the operand loads, instead of as one more
    // dependent round trip behind the barrier
    float ep_sc = 1.f, ep_sh = 0.f;
    if (tid < PG * A) {
        const int co_t = co0 + ((tid % A) / TPO);
        ep_sc = p.scale[co_t < p.N ? co_t : p.N - 1]; ep_sh = p.shift[co_t < p.N ? co_t : p.N - 1];
    }
    float acc[A];
#pragma unroll
    for (int i = 0; i < A; ++i) acc[i] = 0.f;
    // one input channel of this lane: its NP input rows and the taps of the CB output channels (32-bit offsets: the executor
    // bounds every activation below 2^31 elements)
    auto load = [&](int cbase /* first channel of this wave's set: wave-uniform */, float (&x)[NP], f32x4 (&w)[CB], float& cmin) {
        const int c0 = cbase + lane;
        const bool cok = c0 < Ctot;
        const int ci = cok ? c0 : Ctot - 1;
        const float* col;           // source column of this lane's channel (row 0 of window b), and the row pitch
        int Cs;
        if (USRC) {                 // the whole wave reads one source: everything but the lane's channel offset is scalar
            const bool fa = cbase < p.Ca;
            Cs = fa ? p.Ca : p.Cb;
            col = (fa ? p.sa : p.sb) + (size_t)b * (size_t)src_rows * (size_t)Cs + (fa ? 0 : -p.Ca);
        } else {
            const bool fa = ci < p.Ca;
            Cs = fa ? p.Ca : p.Cb;
            col = (fa ? p.sa : p.sb) + (size_t)b * (size_t)src_rows * (size_t)Cs + (fa ? 0 : -p.Ca);
        }
        const unsigned wl = (unsigned)ci * 4u;
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            const int co = co0 + u < p.N ? co0 + u : p.N - 1;                        // scalar
            w[u] = ry_ld4(p.w + ((unsigned)(co * Ctot) * 4u + wl));
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {                                               // out-of-range rows read row 0; `fix` replaces them
            const int pos = pos0 + j;
            const bool ok = pos >= 0 && pos < src_rows;                              // scalar
            x[j] = col[(unsigned)((ok ? pos : 0) * Cs) + (unsigned)ci];
        }
        if (PADMIN) {
            cmin = INFINITY;
            if (need_min) cmin = fminf(fminf(cmins[ci], cmins[64 + ci]), fminf(cmins[128 + ci], cmins[192 + ci]));
        }
    };
    // applied after the scheduling fence, so that no select sits between the loads (the scheduler otherwise waits for the first
    // eight loads before it issues the rest)
    auto fix = [&](int c0, float (&x)[NP], float cmin) {
        const bool cok = c0 < Ctot;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int pos = pos0 + j;
            const bool in = cok && pos >= 0 && pos < p.Lin;
            x[j] = in ? ((PADMIN && pos >= p.n_real) ? cmin : x[j]) : 0.f;
        }
    };
    auto fma_all = [&](const float (&x)[NP], const f32x4 (&w)[CB]) {
#pragma unroll
        for (int u = 0; u < CB; ++u) {
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                if (MODE == RY_C1D_S2) {
                    acc[u * TPO + j] = fmaf(w[u][3], x[2 * j + 3], fmaf(w[u][2], x[2 * j + 2], fmaf(w[u][1], x[2 * j + 1], fmaf(w[u][0], x[2 * j], acc[u * TPO + j]))));
                } else if (MODE == RY_C1D_S1) {
                    acc[u * TPO + j] = fmaf(w[u][3], x[j + 3], fmaf(w[u][2], x[j + 2], fmaf(w[u][1], x[j + 1], fmaf(w[u][0], x[j], acc[u * TPO + j]))));
                } else {        // out[2q] = w1 x[q] + w3 x[q-1];  out[2q+1] = w0 x[q+1] + w2 x[q]   (x[j] = in[r0 - 1 + j])
                    acc[u * TPO + 2 * j] = fmaf(w[u][3], x[j], fmaf(w[u][1], x[j + 1], acc[u * TPO + 2 * j]));
                    acc[u * TPO + 2 * j + 1] = fmaf(w[u][2], x[j + 1], fmaf(w[u][0], x[j + 2], acc[u * TPO + 2 * j + 1]));
                }
            }
        }
    };
    // The channels of a lane are walked up to four at a time: all their loads are in flight before the first FMA, so a layer with
    // 1024 input channels pays ONE memory round trip, not one per 256 channels.
    const int cstep = ktw * 64;
    int cb0 = kw * 64;                                                               // wave-uniform
    constexpr bool QUAD = (NP + 4 * CB) * 4 + A <= 200;                              // register budget of four sets in flight
    if (QUAD) {
        for (; cb0 + 3 * cstep < Ctot; cb0 += 4 * cstep) {
            float x0[NP], x1[NP], x2[NP], x3[NP], m0, m1, m2, m3;
            f32x4 w0[CB], w1[CB], w2[CB], w3[CB];
            load(cb0, x0, w0, m0); load(cb0 + cstep, x1, w1, m1);
            load(cb0 + 2 * cstep, x2, w2, m2); load(cb0 + 3 * cstep, x3, w3, m3);
            ry_sched_fence();                                                        // every load is issued before the first FMA
            fix(cb0 + lane, x0, m0); fix(cb0 + cstep + lane, x1, m1); fix(cb0 + 2 * cstep + lane, x2, m2); fix(cb0 + 3 * cstep + lane, x3, m3);
            fma_all(x0, w0); fma_all(x1, w1); fma_all(x2, w2); fma_all(x3, w3);
        }
    }
    for (; cb0 + cstep < Ctot; cb0 += 2 * cstep) {
        float xa[NP], xb[NP], ma, mb;
        f32x4 wa[CB], wb[CB];
        load(cb0, xa, wa, ma);
        load(cb0 + cstep, xb, wb, mb);
        ry_sched_fence();
        fix(cb0 + lane, xa, ma); fix(cb0 + cstep + lane, xb, mb);
        fma_all(xa, wa);
        fma_all(xb, wb);
    }
    if (cb0 < Ctot) {
        float xa[NP], ma;
        f32x4 wa[CB];
        load(cb0, xa, wa, ma);
        ry_sched_fence();
        RY_OS_STAMP(1)
        fix(cb0 + lane, xa, ma);
        fma_all(xa, wa);
    }
    RY_OS_STAMP(2)
    const float tot = RyReduceScatter64<A, 1>::run(acc, lane);
    RY_OS_STAMP(3)
    if (lane < A) red[wave * 32 + ry_bitrev(lane, LOG2A)] = tot;
    __syncthreads();
    RY_OS_STAMP(4)
    if (tid < PG * A) {
        const int pq = tid / A, idx = tid - pq * A;
        float s = red[(pq * ktw) * 32 + idx];
        for (int k = 1; k < ktw; ++k) s += red[(pq * ktw + k) * 32 + idx];          // fixed order over the ci waves
        const int u = idx / TPO, j = idx - u * TPO;
        const int co = co0 + u;
        const int rg = (tile * PG + pq) * TP;
        const int l = (MODE == RY_C1D_DECONV ? 2 * rg : rg) + j;
        if (co < p.N && l < p.Lout && l < p.keep)
            p.out[((size_t)b * (size_t)p.keep + l) * (size_t)p.N + co] = ry_act(fmaf(s, ep_sc, ep_sh), p.act, p.slope);
    }
    RY_OS_STAMP(5)
#undef RY_OS_STAMP
}

struct RyMaterializeParams {
    RySrc1d s;
    long long npix;             // B*keep: rows written
    int L, keep;                // rows per window in the source / rows kept per window (the crop of the convert wrapper)
    float* out;                 // [B][keep][s.C]
    float slope;
};

RY_KERNEL(256) void ry_materialize(RyMaterializeParams p) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.npix * p.s.C) return;
    const int c = (int)(idx % p.s.C);
    const long long row = idx / p.s.C;
    const size_t pix = (size_t)((row / p.keep) * p.L + row % p.keep);
    p.out[idx] = ry_src1d_load(p.s, pix, c, p.slope);
}

// ---------------------------------------------------------------------------------------------
// Wrapper arithmetic: numpy.pad(mode='minimum') along time, log / drop-last-bin, exp / edge-pad / crop.
// ---------------------------------------------------------------------------------------------
struct RyPadRowsParams {
    const float* in;            // [batch][rows_in][cols_in]
    const float* minv;          // [batch][cols_in] column minima (may be null when rows_out <= rows_in)
    float* out;                 // [batch][rows_out][cols_out]  (cols_out <= cols_in: trailing bins dropped)
    int rows_in, cols_in, rows_out, cols_out;
    int take_log;
    long long in_bstride, out_bstride;
    int minv_bstride;
};

// ry_rep_rows -- rows dst0 .. dst0 + nrows - 1 of every image := row src of the same image (NHWC rows of row_f4 16-byte pieces).  The convert wrapper
// pads a window with copies of ONE row (the column minima), so behind the real frames every encoder layer computes rows that are equal bit for
// bit; the implicit GEMM leaves the tile rows inside that stretch out (RyIgemmParams::hole_ty) and this copy fills them in.
struct RyRepRowsParams { float* base; long long img_f4; int row_f4, src, dst0, nrows; };
RY_KERNEL(256) void ry_rep_rows(RyRepRowsParams p) {     // grid: (pieces of a row / 256, rows to fill, images)
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (i >= p.row_f4) return;
    f32x4* img = reinterpret_cast<f32x4*>(p.base) + (size_t)blockIdx.z * (size_t)p.img_f4;
    img[(size_t)(p.dst0 + (int)blockIdx.y) * p.row_f4 + i] = img[(size_t)p.src * p.row_f4 + i];
}

// ry_pad_min_rows -- column minimum + padded / logged rows in one launch (one graph node on the convert path): a workgroup owns 16
// columns (16 row groups x 16 columns), takes their minimum over the real rows, then writes the padded / logged block.
template <int G>                                              // G row groups x 16 columns = 16 G threads (G = 16 or 64)
RY_KERNEL(16 * G) void ry_pad_min_rows(RyPadRowsParams p) {   // blockIdx.y = window of the batch
    __shared__ float red[16 * G];
    const int tid = (int)threadIdx.x, cl = tid & 15, grp = tid >> 4;
    const int c = (int)blockIdx.x * 16 + cl;
    const float* in = p.in + (size_t)blockIdx.y * (size_t)p.in_bstride;
    float* out = p.out + (size_t)blockIdx.y * (size_t)p.out_bstride;
    const bool cin_ok = c < p.cols_in, cout_ok = c < p.cols_out;
    float m = INFINITY;
    const bool need_min = p.rows_out > p.rows_in;
    // pass 1: rows grp, grp + G, ...: copy (log) the real rows and track the column minimum; 8 loads in flight per lane (with
    // G = 64 a 300-frame window is ONE batch of loads per lane: the kernel is a chain of memory latencies, not bandwidth)
    for (int r0 = grp; r0 < p.rows_in; r0 += 8 * G) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int r = r0 + G * u; v[u] = (cin_ok && r < p.rows_in) ? in[(size_t)r * p.cols_in + c] : INFINITY; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + G * u;
            m = fminf(m, v[u]);
            if (cout_ok && r < p.rows_in && r < p.rows_out) out[(size_t)r * p.cols_out + c] = p.take_log ? logf(v[u]) : v[u];
        }
    }
    if (!need_min) return;
    red[tid] = m;
    __syncthreads();
    float mm = red[cl];
#pragma unroll 8
    for (int g = 1; g < G; ++g) mm = fminf(mm, red[g * 16 + cl]);
    if (!cout_ok) return;
    const float fill = p.take_log ? logf(mm) : mm;
    for (int r = p.rows_in + grp; r < p.rows_out; r += G) out[(size_t)r * p.cols_out + c] = fill;
}

struct RySrPostParams { const float* y; float* out; int rows, cols_in, cols_out; long long y_bstride, out_bstride; };

RY_KERNEL(256) void ry_sr_post(RySrPostParams p) {   // out[r][f] = exp(y[r][min(f, cols_in-1)]), blockIdx.y = window
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)p.rows * p.cols_out) return;
    const int f = (int)(idx % p.cols_out);
    const int r = (int)(idx / p.cols_out);
    const int fi = f < p.cols_in ? f : p.cols_in - 1;
    p.out[(size_t)blockIdx.y * (size_t)p.out_bstride + idx] =
        expf(p.y[(size_t)blockIdx.y * (size_t)p.y_bstride + (size_t)r * p.cols_in + fi]);
}
