// ry_plan.h -- what the three predictor units of libry355.so share: ry_plan.cpp (topology, filter re-layout, the launch planner and its switches),
// ry_exec.cpp (the kernels' launchers, the captured forward, profiling, the single operators) and ry_net.cpp (the C ABI of include/ry355.h).
// Host-side only: nothing here needs ry_kernels.h (one unit, ry_exec.cpp, instantiates the kernels).
#pragma once
#include "ry_dev.h"
#include "ry_host.h"

// ------------------------------------------------------------------------------------------------
// host-side filter re-layout and BatchNormalization folding (once, at creation)
// ------------------------------------------------------------------------------------------------
static const int DECONV_KY[2][2] = {{1, 3}, {0, 2}};   // output parity p, tap t -> kernel index
struct TapTable {
    int nphases = 1, ntaps = 1;
    int dy[4][16], dx[4][16], ky[4][16], kx[4][16], pdy[4], pdx[4];
};

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

// Activation buffers that an implicit-GEMM layer may read end in ZTAIL zeroed floats: the LDS-DMA kernel fetches its padding
// from there (RyConvGeom::zoff1 / zoff2); nothing ever writes them.
// (round 5: a whole zeroed PIXEL of up to 2048 channels -- ry_c2d_os fetches out-of-image taps from it at any channel offset)
static const size_t ZTAIL = 2048;
// bit 0 = input-patch reuse in the deconvolution layers, bit 1 = in the k4 s2 convolution layers (DESIGN.md 5.1 + section 9: A/B measured, both on)
static const int g_patch = 3;

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

// The LDS-DMA pixel path keeps one KiB per (wave, four tile rows, unit in flight): slices with two units in flight and at most 64 KiB of ring
// (what the other window lane's kernels leave free on a CU).
static constexpr bool os2_xl_ok(int mt4, int waves, int depth) { return depth == 2 && mt4 * waves <= 32; }

// ---- what the units share (definitions: ry_plan.cpp / ry_exec.cpp) ----
std::vector<Layer> build_topology(const ry_net_desc& d);
size_t ipow(size_t b, int e);
size_t layer_param_count(const Layer& l, int ndim);
int check_desc(const ry_net_desc* d);
TapTable make_taps(const Layer& l);
float w2d_at(const Layer& l, const float* W, int n, int c, int ky, int kx);
size_t wig_inblock(int nl, int k);
size_t wig16_inblock(int nl, int k);
void relayout_igemm(const Layer& l, const float* W, std::vector<float>& out);
bool wino_eligible(const Layer& l, int ndim);
int prepare_layer(ry_ctx* ctx, Arena& arena, Layer& l, int ndim, float eps, const float* W, const float* b, const float* bn, bool want_os2 = false);
int alloc_ztail(ry_ctx* ctx, Arena& arena, float** p, size_t nfloats);
void tile_dims(int tile, int* bm, int* bn);
extern int g_s2_hole;
extern int g_s2_crop;
extern int g_force[16][3];
extern int g_x3_min_m;
extern int g_autotune;
extern int g_autotune_reps, g_autotune_max;
extern int g_autotune_pick;
const char* tile_name(int tile, int kg, bool bf16, int patch);
int tile_occ(int tile, int kg);
double est_time(long blocks, int bm, int bn, int s, int occ, int kg, int M, int N, int nk);
void choose_igemm(const Layer& l, int M, int nphases, int nk, int* tile, int* splits, int* kg, int bf16 = 0 /* 1 bf16, 2 split-bf16 */);
bool choose_os2(int M, int N, int nphases, int U, int* mt4, int* nt4, int* waves, int* depth, double* cost_out = nullptr);
bool wino_cfg_dims(int cfg, int* wm, int* wn, int* nsl);
void wino_tile_hw(int cfg, int mbw, int* th, int* tw);
const char* wino_name(int cfg, int mode);
bool choose_wino(int Mh, int Mw, int N, int nphases, int npatches, int B, int* cfg, int* mbw, int* splits);
int c1d_mode(const Layer& l);
int c1d_tile_len(int mode);
int choose_splits_1d(const Layer& l, int B, int rows, int mode);
bool plan_tile_rows(const LayerPlan& lp, int Mh, int Mw, int* th, int* tw_out = nullptr);
int build_plan(ry_net* net, Plan& P);
int autotune_plan(ry_net* net, Plan& P);
int get_plan(ry_net* net, int B, int T, int mode, int n_frames, Plan** out);
int run_plan(ry_net* net, Plan& P, const float* x, float* y, int on_device);
int read_plan_env();
int read_env_switches();
unsigned short host_f2bf(float f);
float host_bf2f(unsigned short h);
void build_wigx3(const Layer& l, const std::vector<float>& w32, std::vector<unsigned short>& out);
int profile_plan(ry_net* net, Plan* P, int reps, ry_kernel_stat* stats, int max_stats, int* n_stats);
