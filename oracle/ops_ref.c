/* Plain-C restatement of the Chainer operators on the convert hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED (see oracle/__init__.py): the operators live in `chainer`, called by the un-vendored `yukarin` /
 * `become_yukarin` predictors that the reference reaches from
 * /root/reference/realtime_voice_conversion/yukarin_wrapper/voice_changer.py:33 (stage 1, `AcousticConverter.convert`)
 * and :41 (stage 2, `SuperResolution.convert`).  This file restates their published semantics (SURVEY.md section 8(c)
 * item 1) as direct loop nests -- one output plane at a time, one multiply-add per (output, tap, input channel) -- a third
 * implementation next to the tap-wise tensordot form in ops_numpy.py and torch/oneDNN in torch_ref.py, so that the three can
 * be checked against each other (tests/test_oracle.py).
 *
 * Layout is Chainer's: activations NCHW (1-D layers: H = 1, kh = 1, ph = 0), convolution filters (Cout, Cin, kh, kw),
 * deconvolution filters (Cin, Cout, kh, kw).  Sums run in double when acc64 != 0 (error attribution), else in float like the
 * reference's fp32 arithmetic.  Built by oracle/c_ref.py (gcc -O2 -fopenmp) into oracle/_build/; nothing under
 * realtime_yukarin_amd/ links or loads it. */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>

/* One output plane (fixed b, co) is accumulated tap by tap: for every (ci, ky, kx) the filter scalar is multiplied into the
 * input row segment it touches, so the innermost loop has no bounds tests.  ACC is float (the reference's arithmetic) or
 * double (error attribution). */
#define RY_REF_CONV_PLANE(ACC)                                                                                      \
    for (size_t i = 0; i < (size_t)Ho * Wo; ++i) acc[i] = (ACC)bv;                                                  \
    for (int ci = 0; ci < Cin; ++ci) {                                                                              \
        const float* xp = x + ((size_t)b * Cin + ci) * H * W;                                                       \
        for (int ky = 0; ky < kh; ++ky)                                                                             \
            for (int kx = 0; kx < kw; ++kx) {                                                                       \
                const ACC wv = (ACC)w[(((size_t)co * Cin + ci) * kh + ky) * kw + kx];                               \
                const int off = kx * dw - pw; /* ix = ox * sw + off must lie in [0, W) */                           \
                int lo = off < 0 ? (-off + sw - 1) / sw : 0;                                                        \
                int hi = (W - 1 - off) >= 0 ? (W - 1 - off) / sw : -1;                                              \
                if (hi > Wo - 1) hi = Wo - 1;                                                                       \
                for (int oy = 0; oy < Ho; ++oy) {                                                                   \
                    const int iy = oy * sh - ph + ky * dh;                                                          \
                    if (iy < 0 || iy >= H) continue;                                                                \
                    const float* xr = xp + (size_t)iy * W;                                                          \
                    ACC* ar = acc + (size_t)oy * Wo;                                                                \
                    for (int ox = lo; ox <= hi; ++ox) ar[ox] += wv * (ACC)xr[ox * sw + off];                        \
                }                                                                                                   \
            }                                                                                                       \
    }                                                                                                               \
    for (size_t i = 0; i < (size_t)Ho * Wo; ++i) yp[i] = (float)acc[i];

/* ConvolutionND / Convolution2D: cross-correlation, out = floor((L + 2p - d(k-1) - 1) / s) + 1. */
void ry_ref_conv(const float* x, int B, int Cin, int H, int W, const float* w, const float* bias, int Cout, int kh, int kw,
                 int sh, int sw, int ph, int pw, int dh, int dw, int acc64, float* y) {
    const int Ho = (H + 2 * ph - dh * (kh - 1) - 1) / sh + 1, Wo = (W + 2 * pw - dw * (kw - 1) - 1) / sw + 1;
#pragma omp parallel
    {
        void* buf = malloc((size_t)Ho * Wo * sizeof(double));
#pragma omp for collapse(2) schedule(dynamic)
        for (int b = 0; b < B; ++b)
            for (int co = 0; co < Cout; ++co) {
                float* yp = y + ((size_t)b * Cout + co) * Ho * Wo;
                const float bv = bias ? bias[co] : 0.0f;
                if (acc64) { double* acc = (double*)buf; RY_REF_CONV_PLANE(double) }
                else { float* acc = (float*)buf; RY_REF_CONV_PLANE(float) }
            }
        free(buf);
    }
}

/* Input (iy, ix) of channel ci adds x * w[ci][co][ky][kx] to output (iy * s + ky - p, ix * s + kx - p). */
#define RY_REF_DECONV_PLANE(ACC)                                                                                    \
    for (size_t i = 0; i < (size_t)Ho * Wo; ++i) acc[i] = (ACC)bv;                                                  \
    for (int ci = 0; ci < Cin; ++ci) {                                                                              \
        const float* xp = x + ((size_t)b * Cin + ci) * H * W;                                                       \
        for (int ky = 0; ky < kh; ++ky)                                                                             \
            for (int kx = 0; kx < kw; ++kx) {                                                                       \
                const ACC wv = (ACC)w[(((size_t)ci * Cout + co) * kh + ky) * kw + kx];                              \
                const int off = kx - pw; /* ox = ix * sw + off must lie in [0, Wo) */                               \
                int lo = off < 0 ? (-off + sw - 1) / sw : 0;                                                        \
                int hi = (Wo - 1 - off) >= 0 ? (Wo - 1 - off) / sw : -1;                                            \
                if (hi > W - 1) hi = W - 1;                                                                         \
                for (int iy = 0; iy < H; ++iy) {                                                                    \
                    const int oy = iy * sh + ky - ph;                                                               \
                    if (oy < 0 || oy >= Ho) continue;                                                               \
                    const float* xr = xp + (size_t)iy * W;                                                          \
                    ACC* ar = acc + (size_t)oy * Wo;                                                                \
                    for (int ix = lo; ix <= hi; ++ix) ar[ix * sw + off] += wv * (ACC)xr[ix];                        \
                }                                                                                                   \
            }                                                                                                       \
    }                                                                                                               \
    for (size_t i = 0; i < (size_t)Ho * Wo; ++i) yp[i] = (float)acc[i];

/* DeconvolutionND / Deconvolution2D: transposed convolution, filters (Cin, Cout, kh, kw), out = s(L - 1) + k - 2p. */
void ry_ref_deconv(const float* x, int B, int Cin, int H, int W, const float* w, const float* bias, int Cout, int kh, int kw,
                   int sh, int sw, int ph, int pw, int acc64, float* y) {
    const int Ho = sh * (H - 1) + kh - 2 * ph, Wo = sw * (W - 1) + kw - 2 * pw;
#pragma omp parallel
    {
        void* buf = malloc((size_t)Ho * Wo * sizeof(double));
#pragma omp for collapse(2) schedule(dynamic)
        for (int b = 0; b < B; ++b)
            for (int co = 0; co < Cout; ++co) {
                float* yp = y + ((size_t)b * Cout + co) * Ho * Wo;
                const float bv = bias ? bias[co] : 0.0f;
                if (acc64) { double* acc = (double*)buf; RY_REF_DECONV_PLANE(double) }
                else { float* acc = (float*)buf; RY_REF_DECONV_PLANE(float) }
            }
        free(buf);
    }
}

/* BatchNormalization with fixed statistics (train = False, convert_worker.py:31-32) followed by the activation of the CBR
 * block: act 0 none, 1 leaky_relu(slope), 2 relu.  In place over (B, C, S). */
void ry_ref_bn_act(float* x, int B, int C, size_t S, const float* gamma, const float* beta, const float* mean, const float* var,
                   float eps, int act, float slope) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            float* p = x + ((size_t)b * C + c) * S;
            const float inv = gamma ? 1.0f / sqrtf(var[c] + eps) : 1.0f;
            for (size_t i = 0; i < S; ++i) {
                float v = p[i];
                if (gamma) v = gamma[c] * ((v - mean[c]) * inv) + beta[c];
                if (act == 1) v = v >= 0.0f ? v : v * slope;
                else if (act == 2) v = v > 0.0f ? v : 0.0f;
                p[i] = v;
            }
        }
}
