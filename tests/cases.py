"""Shared parity cases: the same operator / predictor cases run on the emulator (CPU) and on the MI355X."""
import numpy

from oracle import ops_numpy as ops
from oracle import unet

TOL = 1e-4   # BASELINE.json north_star: "within 1e-4 relative fp32"

# B, L, Cin, Cout, k, stride, pad, dilate, transposed, act, splits
CONV1D_CASES = [
    (1, 40, 9, 64, 3, 1, 1, 1, False, 'lrelu', 0),        # encoder c0 shape
    (2, 64, 64, 128, 4, 2, 1, 1, False, 'lrelu', 0),      # encoder down layer, batch 2
    (1, 8, 128, 70, 4, 2, 1, 1, False, 'relu', 4),        # ragged Cout, split over input channels
    (1, 5, 96, 64, 4, 2, 1, 1, True, 'relu', 3),          # decoder up layer, odd length
    (2, 33, 40, 128, 4, 2, 1, 1, True, None, 0),
    (1, 50, 16, 32, 3, 1, 3, 3, False, 'glu', 0),         # dilated conv + GLU (north_star operator coverage)
    (1, 37, 8, 16, 1, 1, 0, 1, False, None, 0),           # 'same' 1x1 layer (extensive_layers < 8)
    (1, 64, 12, 20, 4, 3, 2, 2, False, 'relu', 2),        # generic stride/dilation path
    (1, 1, 16, 64, 4, 2, 1, 1, True, 'relu', 0),          # deepest decoder layer: length 1 -> 2
    (1, 2, 16, 64, 4, 2, 1, 1, False, 'lrelu', 2),        # deepest encoder layer: length 2 -> 1
]

# B, H, W, Cin, Cout, k, stride, pad, transposed, act, path, tile, splits
CONV2D_CASES = [
    (1, 8, 12, 1, 16, 3, 1, 1, False, 'lrelu', 'direct', None, 0),     # SR encoder c0 (Cin = 1)
    (1, 6, 8, 24, 1, 3, 1, 1, False, None, 'direct', None, 0),         # SR decoder c7 (Cout = 1)
    (2, 6, 8, 8, 12, 4, 2, 1, False, 'lrelu', 'direct', None, 0),
    (1, 3, 4, 8, 12, 4, 2, 1, True, 'relu', 'direct', None, 0),
    (1, 12, 16, 32, 128, 4, 2, 1, False, 'lrelu', 'igemm', '32x128', 0),
    (1, 12, 16, 32, 128, 4, 2, 1, False, 'lrelu', 'igemm', '64x128', 2),
    (2, 16, 16, 64, 128, 4, 2, 1, False, 'lrelu', 'igemm', '128x128', 1),
    (1, 16, 36, 32, 64, 4, 2, 1, False, 'relu', 'igemm', '128x64', 3),
    (1, 3, 4, 64, 128, 4, 2, 1, True, 'relu', 'igemm', '32x128', 0),   # M = 12 rows: ragged tile
    (1, 6, 10, 32, 64, 4, 2, 1, True, 'relu', 'igemm', '128x64', 2),
    (1, 5, 7, 64, 128, 3, 1, 1, False, None, 'igemm', '64x128', 0),
    (1, 5, 7, 32, 128, 1, 1, 0, False, 'relu', 'igemm', '32x128', 1),
    (1, 2, 4, 64, 128, 4, 2, 1, False, 'lrelu', 'igemm', None, 0),     # deepest encoder layer, auto tile/split
    (1, 2, 4, 128, 128, 4, 2, 1, False, 'lrelu', 'igemm', '32x128', 21),  # 21 slabs: wide split-K reduce, ragged slab groups
    (2, 7, 10, 1, 64, 3, 1, 1, False, 'lrelu', 'first', None, 0),      # specialised SR encoder c0 (ragged width)
    (2, 5, 9, 128, 1, 3, 1, 1, False, None, 'last', None, 0),          # specialised SR decoder c7 (two-source concat)
    (1, 4, 6, 256, 1, 3, 1, 1, False, None, 'last', None, 0),
    (2, 5, 32, 128, 1, 3, 1, 1, False, None, 'last', None, 0),         # rolling-window form (W % 16 == 0, 128 channels)
    (1, 3, 16, 128, 1, 3, 1, 1, False, None, 'last', None, 0),
    (1, 16, 20, 32, 64, 4, 2, 1, False, 'relu', 'igemm', '128x64', 2),
    (1, 32, 64, 32, 128, 4, 2, 1, False, 'lrelu', 'igemm', '64x128', 1),   # 512 rows: 8 tiles of 64, 2-D 4x16 pixel tiles
    (1, 32, 64, 32, 128, 4, 2, 1, False, 'lrelu', 'igemm', '128x128', 1),  # 16x32 outputs: 2-D 8x16 pixel tiles
    (2, 8, 16, 32, 128, 4, 2, 1, True, 'relu', 'igemm', '64x128', 1),      # deconv, batch 2, 2-D 4x16 tiles over the input grid
    (1, 24, 32, 32, 128, 4, 2, 1, False, 'relu', 'igemm', '96x128', 2),    # 12x16 outputs: 2-D 6x16 tiles, split-K
    (1, 20, 24, 32, 128, 4, 2, 1, False, 'lrelu', 'igemm', '96x128', 1),   # 120 rows: one full 96-row tile + a ragged one
    (1, 6, 8, 64, 256, 4, 2, 1, True, 'relu', 'igemm', '96x128', 2),       # deconv, 2 N-tiles, 4 phases, split-K, XCD-ordered grid
    (1, 6, 10, 64, 64, 4, 2, 1, True, 'relu', 'igemm', '128x64', 0),
    # two K groups per workgroup (split-K summed through the LDS), alone and combined with external split-K
    (1, 24, 32, 64, 128, 4, 2, 1, False, 'relu', 'igemm', '96x128k2', 1),  # 32 chunks: 16 + 16
    (1, 20, 24, 32, 128, 4, 2, 1, False, 'lrelu', 'igemm', '96x128k2', 1), # ragged last tile, 16 chunks
    (1, 16, 16, 96, 128, 3, 1, 1, False, 'lrelu', 'igemm', '128x128k2', 1),# 27 chunks: 13 + 14 (unequal groups)
    (1, 6, 8, 64, 256, 4, 2, 1, True, 'relu', 'igemm', '96x128k2', 2),     # deconv phases, external split 2 x 2 groups (8 chunks -> 2 + 2 | 2 + 2)
    (1, 16, 20, 64, 64, 4, 2, 1, False, 'relu', 'igemm', '128x64k2', 1),
    (2, 8, 16, 32, 128, 4, 2, 1, True, 'relu', 'igemm', '64x128k2', 1),    # 4 chunks per phase: 2 + 2
    (1, 12, 16, 32, 128, 4, 2, 1, False, 'lrelu', 'igemm', '32x128k2', 3), # 16 chunks, 3 external splits of 5/5/6 -> groups of 2+3 / 2+3 / 3+3
    (1, 24, 32, 64, 128, 4, 2, 1, False, 'relu', 'igemm', '96x128k1', 2),  # forced single group
    # sub-pixel deconvolution on 16-pixel-wide 2-D tiles: the input-patch variant (one patch per channel chunk, 4 taps read from it)
    (1, 12, 16, 128, 128, 4, 2, 1, True, 'relu', 'igemm', '96x128', 1),    # 4 chunks x 4 taps, 2 tiles of 6x16 per phase, image borders on every side
    (2, 12, 32, 96, 128, 4, 2, 1, True, 'lrelu', 'igemm', '96x128k2', 1),  # 3 chunks: K groups of 1 + 2 chunks, batch 2, 2 tile columns
    (1, 16, 16, 128, 64, 4, 2, 1, True, 'relu', 'igemm', '128x64', 2),     # 8x16 tiles, external split of 2 + 2 chunks
    (1, 8, 16, 64, 128, 4, 2, 1, True, None, 'igemm', '64x128', 1),        # 4x16 tiles, 2 chunks
    (1, 16, 32, 160, 128, 4, 2, 1, True, 'relu', 'igemm', '128x128k2', 1), # 5 chunks: groups of 2 + 3 (odd counts of 8 / 12 iterations)
    # k4 s2 convolution on 16-pixel-wide 2-D tiles: one patch per (channel chunk, input parity), image borders on every side
    (1, 32, 32, 64, 64, 4, 2, 1, False, 'relu', 'igemm', '128x64', 1),     # 16x16 outputs = 2 tiles of 8x16; 2 chunks x 4 parities
    (2, 16, 32, 96, 128, 4, 2, 1, False, 'lrelu', 'igemm', '64x128k2', 1), # batch 2, 12 patches: K groups of 6 + 6
    (1, 24, 64, 32, 128, 4, 2, 1, False, None, 'igemm', '96x128', 3),      # 2 tile columns, 4 patches over 3 external splits (2 + 1 + 1)
    (1, 32, 32, 160, 128, 4, 2, 1, False, 'relu', 'igemm', '128x128k2', 1),# 20 patches: groups of 10 + 10
]


# ry_c2d_os, the output-stationary weight-streaming kernel on the K-batched v_mfma_f32_4x4x1_16B_f32 (round 5): tile = (mt4, nt4, waves, depth);
# Cin % 128 == 0 cases are handed over as two half-width sources (the un-materialised skip concat)
CONV2D_OS_CASES = [
    (1, 6, 8, 256, 16, 4, 2, 1, False, 'lrelu', 'os', (3, 1, 4, 4), 0),    # 3x4 = 12 output pixels (encoder c7 at 300 frames), image borders on every side; 4 rounds per wave, rotated start
    (1, 12, 16, 512, 32, 4, 2, 1, False, 'lrelu', 'os', (3, 2, 8, 4), 0),  # 48 pixels = four 12-pixel tiles x four 8-channel tiles, two sources of 256 channels, eight waves
    (1, 3, 4, 256, 16, 4, 2, 1, True, 'relu', 'os', (3, 1, 4, 4), 0),      # sub-pixel deconvolution: 4 phases x 12 input pixels (decoder c0), one round per wave
    (2, 3, 4, 512, 24, 4, 2, 1, True, None, 'os', (2, 2, 8, 2), 0),        # batch 2, two sources, eight waves, three 8-pixel tiles, three 8-channel tiles
    (1, 10, 8, 256, 8, 4, 2, 1, False, 'relu', 'os', (4, 2, 4, 4), 0),     # 20 pixels on 16-pixel tiles: a ragged last tile
    (1, 5, 7, 1024, 16, 1, 1, 0, False, 'relu', 'os', (6, 4, 4, 2), 0),    # 'same' 1x1 layer (extensive_layers < 8): 35 pixels on 24-pixel tiles, 96 sums per lane, two sources
    (1, 4, 8, 256, 64, 4, 2, 1, False, 'lrelu', 'os', None, 0),            # the planner's slice
    (1, 2, 4, 256, 128, 4, 2, 1, False, 'lrelu', 'os', (1, 1, 8, 4), 0),   # deepest encoder layer at 100 frames: 1x2 pixels on a 4-pixel tile
    (1, 1, 4, 256, 64, 4, 2, 1, True, 'relu', 'os', (1, 4, 4, 4), 0),      # deepest decoder layer at 100 frames: 4 pixels, 16-channel tiles
    (1, 6, 6, 256, 8, 4, 2, 1, False, None, 'os', (1, 2, 16, 2), 0),       # 9 pixels on 4-pixel tiles, sixteen waves of one round each
    (1, 6, 8, 512, 8, 4, 2, 1, False, 'relu', 'os', (3, 2, 8, 2), 0),   # two units in flight, four rounds per wave, two sources
]


# ry_wino_ldsdma, the k4 s2 p1 layers in Winograd F(2x2, 2x2) form (round 6): tile = (cfg, mbw): cfg 1 = 2 x 2 waves (M-tile of two 8 x 16-pixel blocks,
# 64 channels), cfg 2 = 4 x 2 waves (four blocks); mbw = blocks per tile row.  Cin % 32 == 0 cases are handed over as two half-width sources.
CONV2D_WINO_CASES = [
    (1, 8, 32, 16, 64, 4, 2, 1, True, 'relu', 'wino', (1, 2), 1),      # sub-pixel deconvolution, ONE 8 x 32 tile per phase, one 16-channel patch, single source: image borders on every side
    (1, 16, 16, 32, 64, 4, 2, 1, True, 'relu', 'wino', (1, 1), 1),     # 16 x 16 tile, two sources of 16 channels
    (1, 16, 32, 64, 128, 4, 2, 1, True, 'lrelu', 'wino', (1, 2), 2),   # two tile rows, two channel tiles, external split of 2 + 2 patches (slabs + reduce)
    (1, 32, 64, 32, 64, 4, 2, 1, False, 'lrelu', 'wino', (1, 2), 1),   # k4 s2 convolution: four parity planes per chunk accumulate in the transform domain, 16 x 32 outputs
    (2, 32, 32, 48, 64, 4, 2, 1, False, None, 'wino', (1, 1), 3),      # batch 2, 12 (chunk, parity) patches over 3 splits, single source of 48 channels
    (1, 8, 64, 16, 64, 4, 2, 1, True, 'relu', 'wino', (2, 4), 1),      # eight waves: 8 x 64 tile
    (1, 16, 32, 32, 64, 4, 2, 1, True, 'relu', 'wino', (2, 2), 1),     # 16 x 32 tile
    (1, 32, 16, 64, 128, 4, 2, 1, True, 'lrelu', 'wino', (2, 1), 2),   # 32 x 16 tile, split-K
    (1, 32, 64, 32, 64, 4, 2, 1, False, 'lrelu', 'wino', (2, 2), 1),   # convolution on eight waves
    (2, 64, 32, 48, 64, 4, 2, 1, False, None, 'wino', (2, 1), 3),
    (1, 32, 64, 80, 128, 4, 2, 1, True, 'relu', 'wino', (1, 2), 0),    # 5 patches (odd iteration counts per buffer), the planner's split
    (1, 48, 64, 16, 64, 4, 2, 1, False, 'relu', 'wino', None, 0),      # the planner's everything: 24 x 32 outputs
    (1, 24, 32, 64, 64, 4, 2, 1, True, None, 'wino', (1, 2), 4),       # 3 tile rows, one patch per split
]


def wino_vs_direct(ctx, transposed, seed=41):
    """The Winograd form against the direct implicit GEMM of the same operator on trained-like magnitudes (activations after a ReLU, filters ~ N(0, 0.02)):
    -> (max |y_wino - y_direct| / max |y_direct|, max |y_direct|)"""
    rng = numpy.random.default_rng(seed)
    B, H, W_, Cin, Cout = (1, 16, 32, 128, 64) if transposed else (1, 32, 64, 128, 64)      # a 16 x 32 grid of the stencil either way
    x = numpy.maximum(rng.normal(size=(B, H, W_, Cin)), 0).astype('f4')
    Wt = rng.normal(0, 0.02, size=(Cin, Cout, 4, 4) if transposed else (Cout, Cin, 4, 4)).astype('f4')
    kw = dict(stride=2, pad=1, transposed=transposed, act=None)
    yd = ctx.conv2d(x, Wt, None, None, path='igemm', **kw)
    yw = ctx.conv2d(x, Wt, None, None, path='wino', **kw)
    return float(numpy.abs(yw - yd).max() / numpy.abs(yd).max()), float(numpy.abs(yd).max())


def os_identity_rows(ctx):
    """1x1 'conv' = plain GEMM with one-hot rows on the output-stationary path: pixel i selects input channel 65 i mod 1024, so every K block,
    every K step of a unit and several units are hit.  -> (y [16][16], W rows expected)"""
    Cin, Cout = 1024, 16
    x = numpy.zeros((1, 4, 4, Cin), 'f4')
    for i in range(16):
        x[0, i // 4, i % 4, 65 * i % Cin] = 1.0
    W = (numpy.arange(Cout * Cin, dtype='f4').reshape(Cout, Cin, 1, 1) % 251) / 251.0
    y = ctx.conv2d(x, W, None, None, stride=1, pad=0, path='os', tile=(4, 4, 4, 4))
    return y.reshape(16, Cout), numpy.stack([W[:, 65 * i % Cin, 0, 0] for i in range(16)])      # y[pixel i][n] = W[n][65 i]


def os_every_slice(ctx, one_round=False):
    """Every instantiated slice of ry_c2d_os (tile rows / 4, tile channels / 4, waves, units in flight) on one layer against the implicit GEMM of the
    same operator: a decoder-c1-like sub-pixel deconvolution (48 rows per phase, 64 K units), or with one_round a 1 x 1 layer of 2048 channels whose
    waves run the prologue and the last round only (the pixel value carries its row and its 64-channel chunk: a KiB of a ring slot read before its DMA
    landed shows as a wrong chunk).  -> [(slice, relative error)] of the slices that exist for the shape"""
    rng = numpy.random.default_rng(31)
    if one_round:
        H, W_, Cin, Cout = 3, 8, 2048, 64
        x = ((numpy.arange(H * W_).reshape(H, W_, 1) + 1) + 1000.0 * (numpy.arange(Cin) // 64)[None, None, :]).astype('f4')[None]
        Wt = numpy.ones((Cout, Cin, 1, 1), 'f4')
        kw = dict(stride=1, pad=0, act=None)
    else:
        H, W_, Cin, Cout = 6, 8, 1024, 64
        x = rng.normal(size=(1, H, W_, Cin)).astype('f4')
        Wt = rng.normal(0, 0.02, size=(Cin, Cout, 4, 4)).astype('f4')
        kw = dict(stride=2, pad=1, transposed=True, act=None)
    yi = ctx.conv2d(x, Wt, None, None, path='igemm', **kw)
    out = []
    for n in (1, 2, 4):
        for m in (1, 2, 3, 4, 6):
            for (w, d) in ((4, 4), (8, 4), (8, 2), (16, 2), (4, 2)):
                try:
                    y = ctx.conv2d(x, Wt, None, None, path='os', tile=(m, n, w, d), **kw)
                except RuntimeError as e:
                    assert 'no output-stationary slice' in str(e), e
                    continue
                out.append(((m, n, w, d), float(numpy.abs(y - yi).max() / numpy.abs(yi).max())))
    return out


# 2-D dilated convolution (north_star operator coverage: "1-D/2-D dilated conv"): B, H, W, Cin, Cout, k, stride, pad, dilate, act, path, tile, splits
CONV2D_DILATED_CASES = [
    (1, 12, 16, 32, 128, 3, 1, 2, 2, 'lrelu', 'igemm', '32x128', 0),       # 'same' 3x3 with dilation 2 on the MFMA path
    (2, 10, 14, 64, 128, 3, 1, 3, 3, 'relu', 'igemm', '64x128', 2),        # dilation 3, batch 2, split-K
    (1, 16, 20, 32, 64, 4, 2, 3, 2, None, 'igemm', '128x64', 1),           # k4 s2 with dilation 2 (the gather variant: no input patches)
    (1, 9, 11, 8, 12, 3, 1, 2, 2, 'lrelu', 'direct', None, 0),             # odd channel counts: the direct path
    (1, 7, 9, 5, 6, 2, 1, 0, 4, None, 'direct', None, 0),                  # k2, dilation 4, no padding
]


def run_conv2d_dilated(ctx, rng, case, bn_params):
    B, H, W_, Cin, Cout, k, s, p, d, act, path, tile, splits = case
    x = rng.normal(size=(B, H, W_, Cin)).astype('f4')
    Wt = rng.normal(0, 0.1, size=(Cout, Cin, k, k)).astype('f4')
    b = rng.normal(0, 0.1, Cout).astype('f4')
    bn = bn_params(rng, Cout)
    y = ctx.conv2d(x, Wt, b, bn, stride=s, pad=p, dilate=d, act=act, path=path, tile=tile, splits=splits)
    r = ops.conv_nd(x.transpose(0, 3, 1, 2), Wt, b, stride=s, pad=p, dilate=d)
    r = ops.apply_act(ops.batch_norm_inference(r, *bn), act).transpose(0, 2, 3, 1)
    return y, r


def bf16_round(a):
    """float32 -> nearest-even bfloat16 -> float32 (what v_cvt_pk_bf16_f32 does to the operands of the bf16 kernel)."""
    u = numpy.ascontiguousarray(a, dtype=numpy.float32).view(numpy.uint32).astype(numpy.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.astype(numpy.uint32).view(numpy.float32).reshape(numpy.shape(a))


# B, H, W, Cin, Cout, k, stride, pad, transposed, act, tile, splits   (bf16-operand implicit GEMM, BASELINE config #5)
CONV2D_BF16_CASES = [
    (1, 24, 32, 128, 128, 4, 2, 1, False, 'lrelu', '96x128', 1),           # k4 s2 convolution patch variant in bf16: 2 chunks x 4 parities
    (1, 12, 16, 256, 128, 4, 2, 1, True, 'relu', '96x128', 1),             # deconvolution patch variant in bf16: 4 chunks of 64 channels
    (1, 16, 16, 128, 64, 4, 2, 1, True, 'relu', '128x64k2', 1),            # 2 chunks, one per K group
    (1, 12, 16, 64, 128, 4, 2, 1, False, 'lrelu', '32x128', 0),
    (2, 16, 16, 64, 128, 4, 2, 1, False, 'lrelu', '128x128', 1),
    (1, 32, 64, 64, 128, 4, 2, 1, False, 'relu', '128x128', 2),        # 2-D tiles, split-K
    (1, 6, 10, 128, 64, 4, 2, 1, True, 'relu', '128x64', 0),           # deconv phases
    (1, 24, 32, 64, 128, 4, 2, 1, False, None, '96x128', 1),
    (1, 5, 7, 64, 128, 3, 1, 1, False, None, '64x128', 0),
]


def run_conv2d_bf16(ctx, rng, case, bn_params):
    """-> (y, oracle on bf16-rounded operands, oracle on fp32 operands)"""
    B, H, W_, Cin, Cout, k, s, p, tr, act, tile, splits = case
    x = rng.normal(size=(B, H, W_, Cin)).astype('f4')
    Wt = rng.normal(0, 0.1, size=(Cin, Cout, k, k) if tr else (Cout, Cin, k, k)).astype('f4')
    b = rng.normal(0, 0.1, Cout).astype('f4')
    bn = bn_params(rng, Cout)
    y = ctx.conv2d(x, Wt, b, bn, stride=s, pad=p, transposed=tr, act=act, path='igemm_bf16', tile=tile, splits=splits)
    outs = []
    for xx, ww in ((bf16_round(x), bf16_round(Wt)), (x, Wt)):
        xn = xx.transpose(0, 3, 1, 2)
        r = ops.deconv_nd(xn, ww, b, stride=s, pad=p) if tr else ops.conv_nd(xn, ww, b, stride=s, pad=p)
        outs.append(ops.apply_act(ops.batch_norm_inference(r, *bn), act).transpose(0, 2, 3, 1))
    return y, outs[0], outs[1]


def bf16_split(a):
    """float32 -> (hi, lo) with hi = bf16(a), lo = bf16(a - hi): the operands of the split-bf16 ('bf16x3') mode."""
    hi = bf16_round(a)
    return hi, bf16_round(numpy.asarray(a, dtype=numpy.float32) - hi)


# split-bf16 implicit GEMM (dtype 'bf16x3'): the bf16 cases (K axis three times as long) + odd chunk counts per source
CONV2D_X3_CASES = CONV2D_BF16_CASES + [
    (1, 12, 16, 192, 128, 4, 2, 1, True, 'relu', '96x128k2', 1),           # deconvolution patches: 9 chunks in K groups of 4 + 5
    (1, 16, 32, 64, 128, 4, 2, 1, False, 'lrelu', '64x128', 3),            # convolution patches: 12 (chunk, parity) units over 3 external splits
    (1, 6, 8, 128, 128, 3, 1, 1, False, 'relu', '32x128', 4),              # gather variant, 9 taps x 6 chunks over 4 splits
]


def run_conv2d_x3(ctx, rng, case, bn_params):
    """-> (y, float64 model of the kernel: x_hi w_hi + x_lo w_hi + x_hi w_lo, float64 oracle on the fp32 operands)"""
    B, H, W_, Cin, Cout, k, s, p, tr, act, tile, splits = case
    x = rng.normal(size=(B, H, W_, Cin)).astype('f4')
    Wt = rng.normal(0, 0.1, size=(Cin, Cout, k, k) if tr else (Cout, Cin, k, k)).astype('f4')
    b = rng.normal(0, 0.1, Cout).astype('f4')
    bn = bn_params(rng, Cout)
    y = ctx.conv2d(x, Wt, b, bn, stride=s, pad=p, transposed=tr, act=act, path='igemm_x3', tile=tile, splits=splits)
    f = (lambda xx, ww, bb: ops.deconv_nd(xx, ww, bb, stride=s, pad=p)) if tr else (lambda xx, ww, bb: ops.conv_nd(xx, ww, bb, stride=s, pad=p))
    (xh, xl), (wh, wl) = bf16_split(x), bf16_split(Wt)
    t = lambda a: a.transpose(0, 3, 1, 2).astype('f8')
    zero = numpy.zeros_like(b, dtype='f8')
    r3 = f(t(xh), wh.astype('f8'), b.astype('f8')) + f(t(xl), wh.astype('f8'), zero) + f(t(xh), wl.astype('f8'), zero)
    r = f(t(x), Wt.astype('f8'), b.astype('f8'))
    fin = lambda q: ops.apply_act(ops.batch_norm_inference(q, *bn), act).transpose(0, 2, 3, 1)
    return y, fin(r3), fin(r)


def x3_scaling_property(ctx, rng, case):
    """Size-independent exactness property of the split-bf16 path: scaling the input by a power of two commutes with the bf16
    hi / lo split and with every fp32 accumulation, so y(4 x) == 4 y(x) BIT FOR BIT (no bias, no BN, ReLU), and so does
    scaling the filters.  -> (y(x), y(4 x), y(x; W / 8))"""
    B, H, W_, Cin, Cout, k, s, p, tr, act, tile, splits = case
    x = rng.normal(size=(B, H, W_, Cin)).astype('f4')
    Wt = rng.normal(0, 0.1, size=(Cin, Cout, k, k) if tr else (Cout, Cin, k, k)).astype('f4')
    f = lambda xx, ww: ctx.conv2d(xx, ww, None, None, stride=s, pad=p, transposed=tr, act='relu', path='igemm_x3', tile=tile, splits=splits)
    return f(x, Wt), f(4.0 * x, Wt), f(x, Wt / 8.0)


def run_conv1d(ctx, rng, case, bn_params):
    B, L, Cin, Cout, k, s, p, d, tr, act, splits = case
    x = rng.normal(size=(B, L, Cin)).astype('f4')
    W = rng.normal(0, 0.1, size=(Cin, Cout, k) if tr else (Cout, Cin, k)).astype('f4')
    b = rng.normal(0, 0.1, Cout).astype('f4')
    bn = bn_params(rng, Cout)
    y = ctx.conv1d(x, W, b, bn, stride=s, pad=p, dilate=d, transposed=tr, act=act, splits=splits)
    xn = x.transpose(0, 2, 1)
    r = ops.deconv_nd(xn, W, b, stride=s, pad=p) if tr else ops.conv_nd(xn, W, b, stride=s, pad=p, dilate=d)
    r = ops.apply_act(ops.batch_norm_inference(r, *bn), act).transpose(0, 2, 1)
    return y, r


def run_conv2d(ctx, rng, case, bn_params):
    B, H, W_, Cin, Cout, k, s, p, tr, act, path, tile, splits = case
    x = rng.normal(size=(B, H, W_, Cin)).astype('f4')
    Wt = rng.normal(0, 0.1, size=(Cin, Cout, k, k) if tr else (Cout, Cin, k, k)).astype('f4')
    b = rng.normal(0, 0.1, Cout).astype('f4')
    bn = bn_params(rng, Cout)
    y = ctx.conv2d(x, Wt, b, bn, stride=s, pad=p, transposed=tr, act=act, path=path, tile=tile, splits=splits)
    xn = x.transpose(0, 3, 1, 2)
    r = ops.deconv_nd(xn, Wt, b, stride=s, pad=p) if tr else ops.conv_nd(xn, Wt, b, stride=s, pad=p)
    r = ops.apply_act(ops.batch_norm_inference(r, *bn), act).transpose(0, 2, 3, 1)
    return y, r


def oracle_forward(desc, P, x):
    """channels-last block -> oracle (NC...) -> channels-last."""
    if desc.ndim == 1:
        return unet.unet_forward(x.transpose(0, 2, 1), P, desc.extensive_layers).transpose(0, 2, 1)
    return unet.unet_forward(x[:, numpy.newaxis], P, desc.extensive_layers)[:, 0]


def poisoned_converts(ctx, net, sizes, monkeypatch, modes=('f32', 'bf16', 'bf16x3')):
    """RY_POISON=1 (round 5): every fresh stage-2 activation buffer -- fp32 and bf16 copies -- is filled with NaN patterns, so a kernel that reads a
    row / pixel / channel its producer did not write in THIS forward (dead-row crop, row ranges of a discard, skipped fp32 copies of the bf16
    modes) turns the result into NaN instead of depending on what the allocator handed out.  -> [(n, mode, NaNs, NaNs in the kept rows of a discard)]"""
    import ctypes
    reread = lambda: ctx.reload_env()
    out = []
    try:
        monkeypatch.setenv('RY_POISON', '1'); reread()
        for n, sp in sizes:
            for mode in modes:
                net.set_dtype(mode)                                   # drops the plans: the next convert builds poisoned buffers
                a = net.convert(sp)
                k = max(1, n // 3)
                b = net.convert(sp, discard=(k, k)) if n > 2 * k + 1 else a
                out.append((n, mode, int(numpy.isnan(a).sum()), int(numpy.isnan(b[k:n - k]).sum()) if n > 2 * k + 1 else 0))
    finally:
        monkeypatch.delenv('RY_POISON', raising=False); reread()
        net.set_dtype('f32')
    return out


def wino_properties(ctx, shape, transposed):
    """Size-independent properties of the Winograd operator (`path='wino'`, no activation: the operator is affine in its input) at any layer size:
    affinity y(x1 + x2) + y(0) = y(x1) + y(x2); equivariance -- the input shifted by two rows / columns (one for a transposed convolution) gives the
    output shifted by one (two) away from the borders: every pixel then falls on ANOTHER position of its Winograd tile or on another tile; external
    split-K sums the same products in another order.  Returns the three relative errors (the scale is max |y(x1)|)."""
    B, H, Wd, Cin, Cout = shape
    rng = numpy.random.default_rng(97)
    x1 = rng.normal(size=(B, H, Wd, Cin)).astype('f4'); x2 = rng.normal(size=(B, H, Wd, Cin)).astype('f4')
    Wt = rng.normal(0, 0.05, size=(Cin, Cout, 4, 4) if transposed else (Cout, Cin, 4, 4)).astype('f4')
    b = rng.normal(0, 0.1, Cout).astype('f4')
    kw = dict(stride=2, pad=1, transposed=transposed, act=None, path='wino')
    f = lambda x, **k: ctx.conv2d(x, Wt, b, None, **dict(kw, **k)).astype(numpy.float64)
    y1, y2, y12, y0 = f(x1), f(x2), f(x1 + x2), f(numpy.zeros_like(x1))
    scale = float(numpy.abs(y1).max())
    e_aff = float(numpy.abs(y12 + y0 - y1 - y2).max()) / scale
    si, so = (1, 2) if transposed else (2, 1)                       # input shift -> output shift
    xs = numpy.zeros_like(x1); xs[:, si:, si:] = x1[:, :-si, :-si]
    ys = f(xs)
    m = 2 * so + 2                                                  # margin: pixels whose stencil touches the border or the shifted-in zeros
    e_eq = float(numpy.abs(ys[:, so + m:-m, so + m:-m] - y1[:, m:-so - m, m:-so - m]).max()) / scale
    e_split = float(numpy.abs(f(x1, splits=3) - y1).max()) / scale
    return e_aff, e_eq, e_split
