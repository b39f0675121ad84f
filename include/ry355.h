/* ry355.h -- C ABI of libry355.so: the MI355X-native convert hot path of realtime-yukarin.
 *
 * The reference has no FFI for this path: its boundary is the Python surface of two un-vendored
 * packages.  Each entry point below names the reference call it stands in for; the ctypes binding a
 * maintainer adds is realtime_yukarin_amd/_lib.py.
 *
 * Conventions: plain pointers and sizes, no torch types.  Every function returning int returns 0 on
 * success and a negative RY_E* code on failure (never aborts); ry_last_error() gives the message for
 * the calling thread.  All tensors are float32, channels-last:
 *   stage-1  [batch][frames][channels]      (= the (N, C) feature matrix of encode_feature, untransposed)
 *   stage-2  [batch][frames][bins]
 * Threading: a context and its predictors are used by one thread at a time (the reference converts from a single-threaded worker
 * loop, convert_worker.py:45-59); different contexts / processes are independent.
 * `on_device` = 0: x / y are host pointers, the call returns after the result is in y.
 * `on_device` = 1: x / y are device pointers on the context's GPU; the call only enqueues work on the
 *                  context stream (ry_stream); use ry_sync or your own event to wait.
 */
#ifndef RY355_H
#define RY355_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RY_OK 0
#define RY_EINVAL (-1)   /* bad argument / shape the predictor cannot take */
#define RY_EHIP (-2)     /* HIP runtime error (message carries hipGetErrorString) */
#define RY_ENOMEM (-3)
#define RY_ESTATE (-4)   /* wrong context / destroyed handle */

#define RY_ACT_NONE 0
#define RY_ACT_LRELU 1
#define RY_ACT_RELU 2
#define RY_ACT_GLU 3

typedef struct ry_ctx ry_ctx;
typedef struct ry_net ry_net;

/* `create_predictor(config.model)` / `create_predictor_sr(config.model)` arguments ([MEM], built inside
 * yukarin.AcousticConverter.__init__ / become_yukarin.SuperResolution.__init__; reference call sites
 * realtime_voice_conversion/converter/yukarin_converter.py:40-55, check.py:54-63). */
typedef struct ry_net_desc {
    int ndim;               /* 1 = stage-1 Predictor (Convolution1D), 2 = stage-2 SRPredictor (Convolution2D) */
    int in_ch, out_ch;      /* stage-2: 1, 1 */
    int base;               /* generator_base_channels */
    int extensive_layers;   /* generator_extensive_layers */
    int width;              /* stage-2: bins fed to the predictor (fft_size/2 = 512); stage-1: 1 */
    float bn_eps;           /* Chainer BatchNormalization eps = 2e-5 */
    float lrelu_slope;      /* F.leaky_relu slope = 0.2 */
    int glu;                /* stage-1 only, UNVERIFIED [MEM]: `model.glu_generator` of the stage-1 config -- every conv + BatchNormalization block of
                             * the U-Net produces twice the channels and is gated, a * sigmoid(b), instead of (leaky) ReLU.  The upstream class lives in
                             * the un-vendored `yukarin` package; the strict K-list / shape check of the loader decides whether a real model fits. */
} ry_net_desc;

/* One context per (process, GPU).  Create it in the process that converts (after fork), cf. the
 * convert worker process of run.py:69-79.  `chainer.cuda.get_device(gpu).use()` equivalent. */
int ry_init(int device_ordinal, ry_ctx** out);
void ry_shutdown(ry_ctx* ctx);
int ry_sync(ry_ctx* ctx);
void* ry_stream(ry_ctx* ctx);                 /* hipStream_t the context enqueues on */
int ry_device_count(void);
const char* ry_last_error(void);

/* Number of floats of the flat weight blob = K-list order of the Chainer save_npz keys:
 * encoder/c0/{W,b}, encoder/c{1..7}/{c/W,c/b,batchnorm/gamma,beta,avg_mean,avg_var}, decoder/c{0..6}/..., decoder/c7/{W,b}. */
size_t ry_net_param_count(const ry_net_desc* desc);

/* `chainer.serializers.load_npz(model_path, model)` + `model.to_gpu(gpu)` ([MEM]): takes the flat blob
 * (host, or device when weights_on_device -- e.g. the buffer an RCCL broadcast just filled), folds
 * BatchNormalization into per-channel scale/shift and re-lays the filters out for the kernels. */
int ry_net_create(ry_ctx* ctx, const ry_net_desc* desc, const float* weights, size_t n_floats,
                  int weights_on_device, ry_net** out);
void ry_net_destroy(ry_net* net);
/* A second handle on the same predictor: the filters are shared (freed with the last handle), the clone has its own stream, launch
 * plans, activation buffers and captured graphs.  ry_net_set_dtype on one handle does not change the others. */
int ry_net_clone(ry_net* net, ry_net** out);
/* BASELINE config #5: dtype 1 runs the stage-2 implicit-GEMM layers with bf16 operands on v_mfma_f32_32x32x16_bf16 (fp32 accumulate,
 * bf16 activations between those layers; filters converted once).  dtype 2 ("split-bf16") runs them as three bf16 products per fp32
 * product -- x = hi + lo, w = hi + lo, x w ~ hi hi + lo hi + hi lo -- on the same instruction with fp32 accumulation: results agree
 * with the fp32 path to ~2e-6 (inside the 1e-4 parity bar) at 3/16 of the matrix-pipe time.  dtype 0 (default) is exact fp32.
 * Tolerances: DESIGN.md 5.1 (bf16, split-bf16). */
int ry_net_set_dtype(ry_net* net, int dtype);

/* `Predictor.__call__` / `SRPredictor.__call__` on an already padded block (frames % 128 == 0 when
 * extensive_layers == 8).  stage-1: x [batch][frames][in_ch] -> y [batch][frames][out_ch];
 * stage-2: x [batch][frames][width] -> y [batch][frames][width]. */
int ry_net_forward(ry_net* net, const float* x, float* y, int batch, int frames, int on_device);

/* Array part of `AcousticConverter.convert` (voice_changer.py:33): x [batch][n_frames][in_ch] ->
 * pad time to n + (128 - n % 128) with the per-channel minimum -> Predictor -> crop -> y [batch][n_frames][out_ch]. */
int ry_ac_convert(ry_net* stage1, const float* x, float* y, int batch, int n_frames, int on_device);

/* `SuperResolution.convert` (voice_changer.py:41): sp [batch][n_frames][width+1] -> pad 'minimum' -> log ->
 * drop last bin -> SRPredictor -> edge-pad one bin -> exp -> crop -> out [batch][n_frames][width+1]. */
int ry_sr_convert(ry_net* stage2, const float* sp, float* out, int batch, int n_frames, int on_device);
/* ry_sr_convert for a caller that discards the first `discard_front` / last `discard_back` frames of every window: rows
 * [discard_front, n_frames - discard_back) of `out` are written (bit-identical to ry_sr_convert); the others come back as zeros
 * (host arrays) or are left untouched (device pointers). */
int ry_sr_convert_rows(ry_net* net, const float* sp, float* out, int batch, int n_frames, int discard_front, int discard_back, int on_device);

/* ---- single operators (the Chainer links of SURVEY.md section 2.1), host pointers, Chainer weight layouts ---- */
/* L.ConvolutionND(1) / L.DeconvolutionND(1) [+ L.BatchNormalization] [+ activation].  x [B][L][Cin] ->
 * y [B][Lout][Cout] (GLU: Cout/2 channels).  W: (Cout,Cin,k) or, transposed, (Cin,Cout,k); bn = gamma|beta|avg_mean|avg_var
 * (4*Cout floats) or NULL.  k <= 4; transposed requires k4 s2 p1.  splits = 0 lets the library choose. */
int ry_conv1d(ry_ctx* ctx, const float* x, int B, int L, int Cin, const float* W, const float* bias, const float* bn,
              int Cout, int k, int stride, int pad, int dilate, int transposed, int act, int splits, float* y);
/* L.Convolution2D / L.Deconvolution2D.  x [B][H][W][Cin] -> y [B][Ho][Wo][Cout].  path: 0 auto, 1 implicit-GEMM (MFMA),
 * 2 direct (VALU), 3 SR first layer (1 -> N, 3x3), 4 SR last layer (C -> 1, 3x3, channels read as two sources), 5 implicit-GEMM with bf16 operands,
 * 6 implicit-GEMM in split-bf16 form (x_hi w_hi + x_lo w_hi + x_hi w_lo, fp32 accumulate; the input is split on the host here); tile: 0 auto, 1 = 128x128, 3 = 64x128, 4 = 32x128, 5 = 128x64, 6 = 96x128 (+16: two K groups per workgroup, +32: one). */
int ry_conv2d(ry_ctx* ctx, const float* x, int B, int H, int W, int Cin, const float* Wt, const float* bias, const float* bn,
              int Cout, int k, int stride, int pad, int transposed, int act, int path, int tile, int splits, float* y);
/* the same with a dilation (plain convolution on the implicit-GEMM or the direct path): out = (H + 2 pad - dilate (k - 1) - 1) / stride + 1 */
int ry_conv2d_dilated(ry_ctx* ctx, const float* x, int B, int H, int W, int Cin, const float* Wt, const float* bias, const float* bn,
                      int Cout, int k, int stride, int pad, int dilate, int transposed, int act, int path, int tile, int splits, float* y);

/* ---- the whole device-resident convert: stage-1 -> combine_silent -> mc2sp -> +floor -> stage-2 (voice_changer.py:27-41) ----
 * mc2sp(mc) = exp(mc @ mtx) with mtx [(order+1)][bins] precomputed on the host (every step of pysptk.mc2sp before the exp is
 * linear); see realtime_yukarin_amd/sptk.py.  One H2D and one D2H per window instead of two round trips + a CPU mc2sp. */
typedef struct ry_vc ry_vc;
int ry_vc_create(ry_net* stage1, ry_net* stage2, const float* mtx, int order_plus_1, int bins, ry_vc** out);
void ry_vc_destroy(ry_vc* vc);
/* x_eff [n_eff][in_ch]: encode_feature of the effective (non-silent) frames; row_of[i] = frame index of row i;
 * mc_out [n_frames][order+1] (zeros on silent frames), sp_out [n_frames][bins].  n_eff may be 0 (all silent). */
int ry_vc_convert(ry_vc* vc, const float* x_eff, const int* row_of, int n_eff, int n_frames, float sp_floor,
                  float* mc_out, float* sp_out);
/* The same in two halves, for a convert loop that keeps several windows in flight (the live caller is the worker loop of
 * realtime_voice_conversion/worker/convert_worker.py:45-59: get -> convert -> put).  ry_vc_submit copies the window into a pinned
 * ring slot (six slots), queues H2D -> stage-1 -> combine_silent -> mc2sp -> stage-2 -> D2H on the two predictor streams and returns
 * a ticket WITHOUT waiting; ry_vc_wait blocks for that ticket and copies the results out.  H2D of window i + 1 and D2H of window i - 1
 * run under the kernels of window i.  ry_vc_convert == submit + wait.  RY_ESTATE when all slots are in flight. */
int ry_vc_submit(ry_vc* vc, const float* x_eff, const int* row_of, int n_eff, int n_frames, float sp_floor, int* ticket);
/* The caller will throw away the first `front` and the last `back` frames of every window it gets back -- ConvertStream.process converts
 * buffer + 2 x extra_time and picks the buffer (realtime_voice_conversion/stream/convert_stream.py:40-42).  Stage 2 then computes only the
 * rows that are kept (decoder layers on the row range they depend on; encoder and bottom of the U-Net whole); the kept rows are bit-identical
 * to the full result, the discarded rows of the returned spectrogram are zero (device-pointer calls: left untouched), mc is always complete.
 * Applies to every following ry_vc_submit / ry_vc_submit_wave / ry_vc_enqueue_device(_batch) / ry_vc_stage2_from_mc until changed;
 * (0, 0) = everything. */
int ry_vc_set_discard(ry_vc* vc, int front, int back);
/* Lanes: with `lanes` = 2 or 3 the ring slots run on their own predictor handles (ry_net_clone of the pair given to ry_vc_create: one
 * copy of the filters, separate streams / launch plans / activations), so that the windows in flight execute side by side instead of
 * one stage-2 forward after the other.  Same results.  Device-pointer callers (ry_vc_enqueue_device) must then give windows that are in
 * flight together their own output blocks.  No ticket may be in flight when the lane count changes.  Default 1. */
int ry_vc_set_lanes(ry_vc* vc, int lanes);
int ry_vc_wait(ry_vc* vc, int ticket, float* mc_out, float* sp_out);
/* All pointers on the device, nothing waited for (ry_sync / your own event): consecutive calls pipeline by themselves -- stage-1 of
 * window i + 1 runs on its stream under stage-2 of window i.  bench.py times this. */
int ry_vc_enqueue_device(ry_vc* vc, const float* x_eff_dev, const int* row_of_dev, int n_eff, int n_frames, float sp_floor,
                         float* mc_out_dev, float* sp_out_dev);
/* Several independent windows of n_frames each in one call (streams served side by side; the backlog of run.py's queue): x_eff_dev /
 * row_of_dev hold the effective rows / row maps of the windows one after the other (sum of n_eff[] rows), n_eff is a HOST array of
 * n_windows counts, mc_out_dev [n_windows][n_frames][order+1], sp_out_dev [n_windows][n_frames][bins].  Stage 2 runs as one batch.
 * Window w of the result equals ry_vc_enqueue_device on that window up to summation order (the batch may run under another launch plan). */
int ry_vc_enqueue_device_batch(ry_vc* vc, int n_windows, const float* x_eff_dev, const int* row_of_dev, const int* n_eff, int n_frames,
                               float sp_floor, float* mc_out_dev, float* sp_out_dev);
/* The chain cut where the reference's own class cuts it, so that its unchanged step-by-step calls (voice_changer.py:33-41) keep
 * the data on the device between the two CNNs:
 *   ry_vc_stage1         `acoustic_converter.convert(f_in_effective)`: x_eff [n_eff][in_ch] -> y1_out [n_eff][order+1]; the rows
 *                        also stay on the device;
 *   ry_vc_stage2_from_mc `combine_silent` + `decode_spectrogram` + `sp += floor` + `super_resolution.convert` from those rows:
 *                        sp_out [n_frames][bins]; the intermediate spectrogram never visits the host;
 *   ry_vc_mid_sp         the intermediate spectrogram exp(mc @ mtx) + floor, for a caller that does read it. */
int ry_vc_stage1(ry_vc* vc, const float* x_eff, int n_eff, float* y1_out);
int ry_vc_stage2_from_mc(ry_vc* vc, const int* row_of, int n_eff, int n_frames, float sp_floor, float* sp_out);
int ry_vc_mid_sp(ry_vc* vc, const int* row_of, int n_eff, int n_frames, float sp_floor, float* sp_mid_out);
int ry_vc_reserve_frames(ry_vc* vc, int n_frames);   /* optional: size the ring ahead of the first window */
/* ry_vc_submit with `separate_effective` (voice_changer.py:27-31) ON THE DEVICE: the raw wave (float32) and the feature block of ALL
 * frames go up; frame powers (librosa.feature.rms(center=True, pad 'reflect') ** 2 in float32, numpy's summation order), the gate, the
 * ordered compaction of the effective rows and the scatter back happen there.  The gate is evaluated in the power domain: effective =
 * mse >= p_effective, or every frame when max(mse) >= p_all (power_to_db's top_db clamp); both thresholds come from the host's own
 * float32 log10 by bisection (realtime_yukarin_amd/gate.py), so the mask equals the host formula's bit for bit.  fft_length: a power
 * of two in 128 .. 1024.  ry_vc_wait_wave also returns the mask (n_frames bytes) and the count. */
int ry_vc_submit_wave(ry_vc* vc, const float* wave, int n_samples, int hop, int fft_length, float p_effective, float p_all,
                      const float* feat, int n_frames, float sp_floor, int* ticket);
int ry_vc_wait_wave(ry_vc* vc, int ticket, float* mc_out, float* sp_out, unsigned char* effective_out, int* n_eff_out);
/* The gate alone (`separate_effective`): mask [n_frames], count, and optionally the gathered rows x_eff [n_eff][in_ch] and their
 * frame indices row_of [n_eff] (null to skip). */
int ry_vc_gate(ry_vc* vc, const float* wave, int n_samples, int hop, int fft_length, float p_effective, float p_all,
               const float* feat, int n_frames, unsigned char* effective_out, int* n_eff_out, float* x_eff_out, int* row_of_out);
/* `AcousticConverter.decode_spectrogram` alone (host pointers): sp [n][bins] = exp(mc [n][m] @ mtx [m][bins]) + floor. */
int ry_mc2sp(ry_ctx* ctx, const float* mc, const float* mtx, int n, int m, int bins, float floor, float* sp);

/* ---- chunk parallelism over the GPUs of one node: one RCCL broadcast of each predictor's weight blob from rank 0
 * at start-up, no collective in the steady state; results are re-ordered by window index on the host exactly as run.py:171-183
 * re-orders its `Item.index`.  RCCL is bound at run time (dlopen of librccl.so.1, RY_RCCL_LIB overrides), so the library has no
 * link-time dependency on it and no tensor library is needed: rank 0 calls ry_comm_unique_id and hands the 128 bytes to the other
 * ranks by any host channel (a file, a socket, MPI, a torch.distributed store -- realtime_yukarin_amd/dist.py uses a file next to
 * MASTER_PORT), every rank calls ry_comm_init, ry_dev_alloc, (rank 0: ry_dev_upload,) ry_comm_bcast_weights and
 * ry_net_create(..., weights_on_device = 1). */
typedef struct ry_comm ry_comm;
#define RY_COMM_ID_BYTES 128
int ry_comm_unique_id(void* id128);                                   /* ncclGetUniqueId */
int ry_comm_init(ry_ctx* ctx, const void* id128, int rank, int world, ry_comm** out);   /* ncclCommInitRank on the context's GPU */
void ry_comm_destroy(ry_comm* comm);
int ry_comm_bcast_weights(ry_comm* comm, float* blob_dev, size_t n_floats, int root);  /* in place; returns when it has arrived */
int ry_comm_allreduce_max(ry_comm* comm, double* value);              /* max over the ranks (timing) */
int ry_comm_barrier(ry_comm* comm);
/* device buffers for callers without a tensor library */
int ry_dev_alloc(ry_ctx* ctx, size_t n_floats, float** out);
int ry_dev_free(ry_ctx* ctx, float* p);
int ry_dev_upload(ry_ctx* ctx, float* dst_dev, const float* src_host, size_t n_floats);
int ry_dev_download(ry_ctx* ctx, float* dst_host, const float* src_dev, size_t n_floats);

/* ---- measurement ---- */
int ry_timer_start(ry_ctx* ctx);              /* hipEventRecord on the context stream */
int ry_timer_stop(ry_ctx* ctx, float* ms);    /* record + synchronize + elapsed */

typedef struct ry_kernel_stat {
    char name[48];          /* kernel family as rocprofv3 prints it, e.g. "ry_igemm_ldsdma<96,128,1,4,2,false,1>" */
    char layer[24];         /* e.g. "encoder/c3" */
    float ms;               /* average duration over `reps` launches (hipEvents on the context stream) */
    double flops;           /* algorithmic FLOPs of this launch */
    double bytes;           /* algorithmic bytes: weights + input + output once */
    int grid[3];
    double flops_exec;      /* matrix-pipe FLOPs the launch executes: = flops, except 9 / 16 of it for the Winograd F(2x2, 2x2) kernels */
} ry_kernel_stat;
/* Runs the forward `reps` times launch by launch, bracketing every kernel with HIP events. */
int ry_net_profile(ry_net* net, int batch, int frames, int reps, ry_kernel_stat* stats, int max_stats, int* n_stats);
/* The same for the convert wrapper on ONE window of n_frames (what ry_ac_convert / ry_sr_convert / the ry_vc_* window call run:
 * pad kernel, layers, fused crop; stage 2 skips the decoder rows that only feed the padding the wrapper throws away). */
int ry_net_profile_window(ry_net* net, int n_frames, int reps, ry_kernel_stat* stats, int max_stats, int* n_stats);

/* diagnostics: ratio[i * n + j] = wall time of a `us`-microsecond spin kernel on each of two fresh streams i and j, divided by `us`:
 * ~1 when the two streams run side by side, ~2 when one waits for the other (scripts/gpu_queues.py). */
int ry_debug_stream_overlap(ry_ctx* ctx, int n, int us, float* ratio);

/* diagnostics / tests: read the process-wide RY_* environment switches again (otherwise read by ry_init).  Launch plans
 * that exist keep their choices until ry_net_set_dtype drops them. */
int ry_debug_reload_env(void);

/* diagnostics: the launch configuration the stage-2 planner picks for an implicit-GEMM layer with M output rows (pixels of
 * one sub-pixel phase), Cout output channels, `nphases` phases (4 for the k4s2 deconvolution, else 1) and K = 32 * nk:
 * tile code (see ry_conv2d), external split-K count, K groups per workgroup, estimated microseconds.  No device work. */
int ry_debug_plan_igemm(int M, int Cout, int nphases, int nk, int* tile, int* splits, int* kgroups, double* est_us);
/* the same for the bf16 (mode 1) / split-bf16 (mode 2) kernels (nk in 64-channel chunks; non-zero tile / splits / kgroups on entry are kept) */
int ry_debug_plan_igemm_bf16(int mode, int M, int Cout, int nphases, int nk, int* tile, int* splits, int* kgroups, double* est_us);

/* diagnostics: the slice the planner picks for an output-stationary layer (ry_c2d_os, round 5) with M rows per phase (batch x pixels), Cout
 * output channels, `nphases` phases and K = 64 * units: tile rows / 4, tile channels / 4, waves per workgroup, units in flight per wave, and the
 * slice cost (x units: what is compared with RY_OS2_MAXCOST to decide between this kernel and the implicit GEMM).  Non-zero values on entry
 * are kept.  Returns RY_EINVAL when no slice fits the shape.  No device work. */
int ry_debug_plan_os2(int M, int Cout, int nphases, int units, int* mt4, int* nt4, int* waves, int* depth, double* cost);

/* diagnostics: the plan the planner picks for a k4 s2 p1 stage-2 layer in Winograd F(2x2, 2x2) form (ry_wino_ldsdma, round 6) whose stencil output grid is
 * Mh x Mw pixels per phase and window, with Cout output channels, `nphases` phases (4: transposed convolution, 1: convolution), K = 16 * npatches input
 * channels (x 4 parities for a convolution: count them in npatches) and `batch` windows: workgroup shape (1 = 2 x 2 waves, 2 = 4 x 2 waves), M-blocks of
 * 8 x 16 pixels per tile row, external split-K.  Non-zero values on entry are kept.  Returns RY_EINVAL when no tile of a shape divides the grid (the layer
 * then stays on the direct implicit GEMM).  No device work. */
int ry_debug_plan_wino(int Mh, int Mw, int Cout, int nphases, int npatches, int batch, int* cfg, int* mbw, int* splits);

#ifdef __cplusplus
}
#endif
#endif
