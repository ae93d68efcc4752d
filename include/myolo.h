/* libmyolo -- C ABI of the MI355X (gfx950) kernels behind the multiyolov5 hot path.
 *
 * The reference (TomMao23/multiyolov5) is pure Python/PyTorch and has NO plugin/FFI/operator registry
 * (SURVEY.md §8b): its "operators" are the ATen calls made by models/common.py, models/yolo.py,
 * utils/loss.py and utils/general.py.  Each entry point below replaces the ATen call(s) cited next to it.
 * The host-side mirror of the reference's module API (multiyolov5_amd/models, multiyolov5_amd/utils)
 * binds these symbols through ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions (all entry points):
 *   - plain device pointers + sizes; the caller owns every buffer; nothing is allocated, freed or
 *     synchronised inside; work is enqueued on `stream` (a hipStream_t) and is hipGraph-capturable;
 *   - return value is a hipError_t as int (0 = ok); MYOLO_EINVAL (-22) for a rejected argument;
 *   - activations are NHWC *views*: element (n,y,x,c) lives at ptr + n*sn + y*sh + x*sw + c
 *     (strides in elements).  A channel slice of a wider buffer (concat-free writes) is just a view
 *     with sw > c.  16-byte alignment of ptr/strides is required wherever a tensor is a conv input;
 *   - dtype is the storage type of activations / packed weights (MYOLO_F16 or MYOLO_F32);
 *     accumulation, BN statistics, losses and gradients of parameters are always fp32.
 */
#ifndef MYOLO_H_
#define MYOLO_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MYOLO_F32 0
#define MYOLO_F16 1
#define MYOLO_U8  2
#define MYOLO_I64 3

#define MYOLO_ACT_NONE    0
#define MYOLO_ACT_SILU    1
#define MYOLO_ACT_SIGMOID 2

#define MYOLO_EINVAL (-22)
#define MYOLO_MAX_TAPS 25
#define MYOLO_STAT_COPIES 32

typedef struct myolo_tensor {
  void*   ptr;
  int32_t n, h, w, c;
  int64_t sn, sh, sw;      /* strides in elements; channel stride is 1 */
  int32_t dtype;
  int32_t reserved;
} myolo_tensor;

/* ---- version / capability ------------------------------------------------------------------ */
int myolo_version(void);                 /* ABI version of this header */
const char* myolo_arch(void);            /* "gfx950" */

/* tuning / test knobs: "stream_min_tiles" (minimum number of 32-pixel row tiles for the streaming conv kernel, default 2048; the
 * parity tests set 1 so that small shapes exercise it), "stream_off" (1: always use the LDS-tiled kernel), "halo_min_tiles" (minimum
 * number of output tiles for the LDS-halo k x k kernel, default 96; tests set 1), "halo_off" (1: never use it).  Returns 0 or
 * MYOLO_EINVAL. */
int myolo_set_option(const char* name, int value);

/* ---- weights ------------------------------------------------------------------------------- */
/* OIHW master weights (nn.Conv2d.weight, common.py:38) -> MFMA-friendly packed layout
 *   transpose=0: dst[rows=cout_pad][ntaps][cols=cin_pad]   (forward operand, K = tap*cin contiguous)
 *   transpose=1: dst[rows=cin_pad ][ntaps][cols=cout_pad]  (dgrad operand)
 * zero padded.  `row_scale` (optional, fp32[cout]) multiplies output-channel rows: folds a BN
 * scale into the weights exactly like fuse_conv_and_bn (utils/torch_utils.py:193-195). */
int myolo_pack_weight(const void* w_oihw, int src_dtype, int cout, int cin, int kh, int kw,
                      void* dst, int dst_dtype, int rows_pad, int cols_pad, int transpose,
                      const float* row_scale, void* stream);

/* all weight packs of a plan in one launch.  jobs: device int64 [njobs][12] = {src, dst, cout, cin, ntaps, rows_pad,
 * cols_pad, transpose, src_dtype, dst_dtype, src2, cout2}; chunks: device int32 [nchunks][2] = {job, first packed element}.
 * src2 (0 = none): a second OIHW tensor [cout2][cin][taps] of the same dtype stacked behind src along cout.
 * chunk_elems <= 0: LDS-tiled mode (-chunk_elems = the largest tap count of any job, 0 = up to MYOLO_MAX_TAPS: sizes the LDS
 * tile) -- chunks = {job, tile}; a tile is 16 co x 32 ci (transpose = 0) or 32 co x 16 ci (transpose = 1)
 * over all taps (64 co x 64 ci for 1x1 weights), tiles numbered co-tile major over ceil(cout_all / tco) x ceil(cin / tci); only the valid region is written (the
 * caller keeps the padding of dst zero). */
int myolo_pack_weights_mt(const int64_t* jobs, const int32_t* chunks, int nchunks, int chunk_elems, void* stream);

/* Focus slicing + cat (common.py:550) fused with the image cast: NCHW [n,3,h,w] (f32|f16|u8, value*mul)
 * -> NHWC [n,h/2,w/2,16]: channel 3*q+c, q=(row parity, col parity) in order (0,0),(1,0),(0,1),(1,1);
 * channels 12..15 are zero. */
int myolo_focus_pack(const void* img_nchw, int src_dtype, int n, int h, int w, float mul,
                     const myolo_tensor* out, void* stream);

/* ---- convolution (implicit GEMM on MFMA) ------------------------------------------------------
 * Replaces nn.Conv2d (+ BatchNorm2d + SiLU) of `Conv` (common.py:34-46), the bare dilated
 * Conv2d+BN+SiLU triples (common.py:481-490), Detect.m[i] (yolo.py:211-214) and, with transposed
 * weights / flipped taps, the autograd dgrad of all of them.
 *   y[n,oy,ox,:] (+)= epilogue( sum_t  x[n, (oy*stride+tap_dy[t])>>up, (ox*stride+tap_dx[t])>>up, :] . W[:,tap_w[t],:] )
 *   epilogue(v) = act(v*scale[c] + shift[c]) + res      (each part optional)
 *   stats != NULL: atomically accumulates per-channel sum / sum of squares of the raw fp32 accumulators into ONE of
 *                  MYOLO_STAT_COPIES (32) interleaved copies [copy][2][cout] (copy = workgroup id % 32; the consumer adds the
 *                  copies): stats is fp32[8*2*cout], zeroed by the caller   (training-mode BatchNorm statistics)
 *   det_no  > 0 : y is written in Detect's permuted layout [n, na, h, w, det_no] (yolo.py:214)           */
/* BatchNorm-backward statistics produced by the launch that COMPLETES an activation gradient (the reduce pass of
 * myolo_bn_act_bwd_reduce folded into the epilogue of the dgrad that writes the last contribution to gout):
 *   for output channels [c0, c1) of the launch:  dsum[copy][0..C) += sum dz,  dsum[copy][C..2C) += sum dz*xhat,   C = c1 - c0,
 *   dz = gout * act'(z), z = (y - mean)*invstd*gamma + beta, xhat = (y - mean)*invstd, gout = the value the launch stores (after
 *   `accumulate`, rounded to the storage type), y = raw conv output of the normalised layer at the same pixel.
 * Segments whose bounds are not multiples of the kernel's N tile (or launches served by a kernel without this epilogue) are handled
 * by an internal myolo_bn_act_bwd_reduce launch over the stored gout: the caller gets `dsum` either way. */
#define MYOLO_MAX_BNB 4
typedef struct myolo_bn_bwd_seg {
  int32_t c0, c1;            /* channel range inside the launch's output y */
  myolo_tensor y;            /* raw conv output of the normalised layer: [N,Ho,Wo,c1-c0], same pixels as the launch's y */
  const float* saved;        /* fp32[2*C]: mean, invstd (myolo_bn_act_fwd) */
  const float* gamma;
  const float* beta;
  float*  dsum;              /* fp32[MYOLO_STAT_COPIES*2*C], zeroed by the caller */
  int32_t act;
  int32_t reserved;
} myolo_bn_bwd_seg;

typedef struct myolo_conv_desc {
  myolo_tensor x;            /* [N,Hi,Wi,Cin] (source dims; logical dims are <<up_shift) */
  myolo_tensor y;            /* [N,Ho,Wo,Cout] */
  const void*  w;            /* packed weights [cout_pad][wtaps][cin_pad], dtype == x.dtype */
  int32_t cin_pad, cout_pad, wtaps;
  int32_t ntaps, stride, up_shift;
  int32_t tap_dy[MYOLO_MAX_TAPS], tap_dx[MYOLO_MAX_TAPS], tap_w[MYOLO_MAX_TAPS];
  const float* scale;        /* fp32[cout] or NULL */
  const float* shift;        /* fp32[cout] or NULL */
  int32_t act;
  int32_t accumulate;        /* 1: y += result */
  myolo_tensor res;          /* res.ptr == NULL: none */
  float*  stats;             /* fp32[MYOLO_STAT_COPIES*2*cout] or NULL */
  int32_t det_no;
  int32_t nbnb;              /* number of entries of `bnb` (0..MYOLO_MAX_BNB) */
  const myolo_bn_bwd_seg* bnb;   /* BatchNorm-backward statistics to produce for (slices of) y, or NULL */
} myolo_conv_desc;
int myolo_conv(const myolo_conv_desc* d, void* stream);

/* dgrad of a 1x1 stride-1 Conv + BatchNorm (+ activation) layer WITH the BatchNorm-backward apply pass in its operand path (round 4;
 * reference models/common.py:42-43 `act(bn(conv(x)))`, autograd): d->x is the gradient w.r.t. the layer's ACTIVATION output (gout), `f`
 * names the layer's saved raw conv output y, the statistics of the reduce pass and where dy (the gradient w.r.t. y, the weight gradient's
 * operand) goes.  Same results as myolo_bn_act_bwd_apply(gout, y, ..., dy, no gres) followed by myolo_conv(d with x = dy) -- which is what
 * runs when the layer does not qualify -- in one launch: gout and y are staged once, dy = sc*dz + cb*y + cd is formed in LDS, feeds the
 * MFMAs and is written out by the workgroups of the first N tile; dgamma / dbeta += the reduce pass' sums. */
typedef struct myolo_bn_apply_fold {
  myolo_tensor y;            /* raw conv output of the layer [N,H,W,C] (saved by the forward) */
  myolo_tensor dy;           /* OUT [N,H,W,C]: gradient w.r.t. y */
  const float* saved;        /* fp32[2*C]: mean, invstd (myolo_bn_act_fwd) */
  const float* gamma;
  const float* beta;
  const float* dsum;         /* fp32[MYOLO_STAT_COPIES*2*C]: sums of myolo_bn_act_bwd_reduce / myolo_conv_desc.bnb */
  float* dgamma;             /* += (may be NULL) */
  float* dbeta;
  int32_t act;
  int32_t reserved;
} myolo_bn_apply_fold;
int myolo_conv_dgrad_bn(const myolo_conv_desc* d, const myolo_bn_apply_fold* f, void* stream);

/* Round 6 -- north_star's "fused Conv+BN+SiLU" in TRAINING mode: myolo_conv(d) (raw output + batch statistics, d->stats != NULL) and
 * myolo_bn_act_fwd_split(d->y, d->stats, ...) in ONE launch: every output tile of the layer is resident with its accumulators held, the
 * per-channel sums cross a device-wide barrier inside the launch (csrc/myolo_dev.h grid_barrier_xcd), then each workgroup stores the raw
 * output (the backward needs it) AND out = act((y - mean) * invstd * gamma + beta) (+ res) from the same registers.  Results as the two
 * launches (reference models/common.py:42-43 `self.act(self.bn(self.conv(x)))`, Bottleneck add common.py:105): y rounded to the storage type
 * before it is normalised, statistics from the fp32 accumulators, `saved` / running statistics / num_batches_tracked updated.
 * `barrier`: MYOLO_GRID_BARRIER_BYTES, zeroed once (see myolo_bn_act_bwd_fused).  myolo_conv_bn_act_ok(d): the fused kernel will run (fp16,
 * conv_mid's layer classes, at most one 8-wave tile per CU: the 32x64 / 16x32 maps at batch 16); otherwise myolo_conv_bn_act runs the two
 * launches, which is also the definition of the result. */
typedef struct myolo_bn_fwd_fuse {
  const float* gamma;            /* first parameter set (channels [0, split->c_split) when split != NULL) */
  const float* beta;
  float* running_mean;           /* may be NULL */
  float* running_var;
  int64_t* nbt;
  float* saved;                  /* fp32[2*cout]: mean, invstd */
  float eps, momentum;
  int32_t act;
  int32_t reserved;
  myolo_tensor res;              /* added AFTER the activation (ptr NULL: none) */
  myolo_tensor out;
  const struct myolo_bn_split* split;   /* second parameter set (merged cv1 | cv2; declared below), may be NULL */
  uint32_t* barrier;
} myolo_bn_fwd_fuse;
int myolo_conv_bn_act_ok(const myolo_conv_desc* d);
int myolo_conv_bn_act(const myolo_conv_desc* d, const myolo_bn_fwd_fuse* f, void* stream);

/* Two convolutions in one launch, eval epilogues (round 5): y_b = conv_b(conv_a(x)) where `a` is a 1x1 stride-1 layer whose output
 * (a->y == b->x, the same view) has NO other reader and `b` a 3x3 stride-1 dilation-1 layer over the same map -- the fused model's
 * Bottleneck (reference models/common.py:95-105 `x + cv2(cv1(x))` with Conv.fuseforward, common.py:45-46; the shortcut is b->res).  Each
 * descriptor keeps the meaning it has for myolo_conv (scale / shift / act / res); the intermediate is rounded to the storage type where
 * the two-launch form stores it.  When the fused kernel runs (fp16, 64 / 128 / 256 channels in, mid and out ... csrc/conv_pair.hip) a->y
 * is NOT written; every other pair runs as myolo_conv(a) followed by myolo_conv(b), which is also the definition of the result.
 * Aliasing: b->y may overlap a->x or b->res (an in-place Bottleneck) -- such a pair is never fused (neighbouring workgroups would still
 * read the input halo while others store the output); it runs as the two launches, for which that aliasing is harmless. */
int myolo_conv_pair(const myolo_conv_desc* a, const myolo_conv_desc* b, void* stream);

/* dgrad of a STRIDE-2 convolution.  `parity[k]`, k = 2*py + px, is the stride-1 sub-convolution producing the input-gradient pixels
 * (2a+py, 2b+px) from dy and the transposed weights (the taps with (p + pad - k*d) % 2 == 0; y = the strided parity view of gx).  With
 * all n == 4 parities over one dy / one weight tensor the four run as ONE launch (dy staged once, every gx row written whole);
 * otherwise, or when the layer does not qualify (fp32, weight panel too large), each is a myolo_conv. */
int myolo_conv_dgrad_s2(const myolo_conv_desc* const* parity, int n, void* stream);

/* wgrad: dw_oihw[co][ci][t] += sum_{n,oy,ox} dy[n,oy,ox,co] * x[n, (oy*stride+tap_dy[t])>>up, (ox*stride+tap_dx[t])>>up, ci]
 * (into an OIHW fp32 gradient buffer = Parameter.grad layout; split-K partials go through `ws` when given, else fp32 atomics).
 * db (optional, fp32[cout]) += sum dy. */
typedef struct myolo_wgrad_desc {
  myolo_tensor x, dy;
  float*  dw;
  float*  db;
  int32_t ntaps, stride, up_shift;
  int32_t tap_dy[MYOLO_MAX_TAPS], tap_dx[MYOLO_MAX_TAPS];
  int32_t ksplit;            /* 0 = auto */
  int32_t cout, cin;         /* real weight dims when dy.c / x.c are channel-padded views (0: use dy.c / x.c) */
  int32_t wg_hint;           /* 0: library default; > 0: target number of workgroups for this launch (split-K = wg_hint / output blocks).
                                The caller knows how much dependent work is still queued behind this gradient: few long-lived
                                workgroups while there is, many for the last layers of a backward pass (their latency is exposed) */
  float*  ws;                /* optional split-K workspace (16-byte aligned): partial tiles are stored there and summed by a
                                second launch instead of fp32 atomics into dw; NULL: atomics */
  int64_t ws_bytes;
} myolo_wgrad_desc;
int myolo_conv_wgrad(const myolo_wgrad_desc* d, void* stream);

/* Round 6: the STEM layer's BatchNorm backward and weight gradient in one pass (csrc/stem_wgrad.hip).  A Conv + BatchNorm + SiLU layer whose
 * input needs no gradient (the network's first layer, reference models/common.py:540-551 Focus.conv) has ONE reader of its dy: its own weight
 * gradient.  dy is linear in sums that need no second pass over the tensors -- dW = sc * (sum g(x)x - k0 * sum x - k1 * sum xhat(x)x), g = gout *
 * silu'(z), xhat = (y - mean) * invstd, k0 / k1 = the BatchNorm-backward means -- so bn_act_bwd_reduce + bn_act_bwd_apply + myolo_conv_wgrad
 * (three launches over the step's largest tensors, at the very end of the backward) become one pass over gout, y, x plus two tiny launches:
 *   dw (+)= autograd's weight gradient of conv -> BatchNorm(train) -> SiLU for the output gradient `gout`;  dgamma += sum g * xhat;  dbeta += sum g.
 * d: x, dw, the nine taps, cout / cin, ksplit (0: 768 workgroups); d->dy is NOT read (dy never exists), d->db must be NULL.  3x3, stride 1,
 * fp16, <= 32 output and <= 16 input channels, map height % 8 == 0 and width % 16 == 0 (myolo_bn_wgrad_stem_ok); `ws`: caller-owned scratch of
 * myolo_bn_wgrad_stem_ws_bytes() bytes that no other launch in flight uses. */
int myolo_bn_wgrad_stem_ok(const myolo_wgrad_desc* d, const myolo_tensor* gout);
int64_t myolo_bn_wgrad_stem_ws_bytes(void);
int myolo_bn_wgrad_stem(const myolo_wgrad_desc* d, const myolo_tensor* gout, const myolo_tensor* y, const float* saved, const float* gamma,
                        const float* beta, int act, float* dgamma, float* dbeta, float* ws, int64_t ws_bytes, void* stream);

/* ---- BatchNorm(+SiLU)(+residual) around a raw conv output (training mode) ----------------------
 * fwd: nn.BatchNorm2d batch-stat path + nn.SiLU + Bottleneck add (common.py:43,105), eps/momentum from
 * initialize_weights (torch_utils.py:150-151).  `stats` are the sums produced by myolo_conv (MYOLO_STAT_COPIES copies).
 *   mean = s/M, var = q/M - mean^2;  out = act((y-mean)*rsqrt(var+eps)*gamma + beta) + res
 *   block 0 also: saved[0..c)=mean, saved[c..2c)=invstd; running_mean/var (unbiased) momentum update,
 *   num_batches_tracked += 1 (int64).
 * gamma == NULL: no normalisation (plain activation of a BN-less conv, e.g. FFM's SE convs). */
int myolo_bn_act_fwd(const myolo_tensor* y, const float* stats, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, int64_t* num_batches_tracked,
                     float* saved, float eps, float momentum, int act,
                     const myolo_tensor* res, const myolo_tensor* out, void* stream);
/* bwd pass 1: dsum[copy][0..c) += sum dz, dsum[copy][c..2c) += sum dz*xhat   with dz = gout * act'(z); dsum is
 * fp32[MYOLO_STAT_COPIES*2*c] zeroed by the caller, pass 2 adds the copies */
int myolo_bn_act_bwd_reduce(const myolo_tensor* gout, const myolo_tensor* y, const float* saved,
                            const float* gamma, const float* beta, int act, float* dsum, void* stream);
/* bwd pass 2: dy = gamma*invstd*(dz - dsum0/M - xhat*dsum1/M); dgamma += dsum1, dbeta += dsum0 (block 0);
 * gres (optional) (+)= gout  (residual branch of Bottleneck) */
int myolo_bn_act_bwd_apply(const myolo_tensor* gout, const myolo_tensor* y, const float* saved,
                           const float* gamma, const float* beta, int act, const float* dsum,
                           float* dgamma, float* dbeta, const myolo_tensor* dy,
                           const myolo_tensor* gres, int gres_accumulate, void* stream);

/* The same three passes over the output of ONE convolution whose channels [0, c_split) and [c_split, c) belong to two BatchNorm
 * modules (C3.cv2 | C3.cv1 run as one launch on their common input, common.py:137): the first parameter set (gamma, beta, running_*,
 * nbt, dgamma, dbeta) serves the low channels, `split` the high ones; stats / saved / dsum stay one array over all c channels.
 * split == NULL: identical to the plain entry points. */
typedef struct myolo_bn_split {
  int32_t c_split;               /* multiple of the 16-byte vector; == c: no second parameter set (gamma2 .. dbeta2 unused) */
  int32_t count_scale;           /* 0 / 1: the statistics cover this tensor's n*h*w samples.  w > 1: `stats` / `dsum` hold the SUMS over w
                                  * ranks with equally sized batches (nn.SyncBatchNorm, train.py:190-193: the caller all-reduced the
                                  * arrays between the producing launch and this one): mean / variance / the unbiased running variance and
                                  * the backward's mean terms use w*n*h*w samples; dgamma / dbeta receive sum / w (= the rank-local sums
                                  * averaged over the ranks, what DistributedDataParallel makes of SyncBatchNorm's local weight gradients) */
  const float* gamma2;
  const float* beta2;
  float* running_mean2;
  float* running_var2;
  int64_t* nbt2;
  float* dgamma2;
  float* dbeta2;
} myolo_bn_split;
/* Round 6: bn_act_bwd_reduce + bn_act_bwd_apply in ONE launch for tensors the resident grid holds in registers (gout and y are read
 * once; partial sums -> device-wide barrier inside the launch -> dx from the same registers).  Same arguments and results as the two
 * entry points it replaces (reference: autograd of nn.BatchNorm2d + nn.SiLU, models/common.py:42-43); `dsum` zeroed by the caller;
 * `barrier`: MYOLO_GRID_BARRIER_BYTES of device memory, 128-byte aligned, zeroed ONCE (the barrier resets itself; launches that may be
 * in flight at the same time need separate blocks; word 18*32 is a sticky timeout flag: a spin that gave up).  Tensors that do not fit
 * the resident grid (myolo_bn_act_bwd_fused_ok == 0), strided views and MYOLO_BN_BWD_FUSED=0 run the two launches. */
#define MYOLO_GRID_BARRIER_BYTES (19 * 32 * 4)
int myolo_bn_act_bwd_fused_ok(int dtype, int64_t n_pixels, int c);
int myolo_bn_act_bwd_fused(const myolo_tensor* gout, const myolo_tensor* y, const float* saved, const float* gamma,
                           const float* beta, int act, float* dsum, float* dgamma, float* dbeta, const myolo_tensor* dy,
                           const myolo_tensor* gres, int gres_accumulate, const myolo_bn_split* split, uint32_t* barrier,
                           void* stream);
int myolo_bn_act_fwd_split(const myolo_tensor* y, const float* stats, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, int64_t* num_batches_tracked,
                           float* saved, float eps, float momentum, int act,
                           const myolo_tensor* res, const myolo_tensor* out, const myolo_bn_split* split, void* stream);
int myolo_bn_act_bwd_reduce_split(const myolo_tensor* gout, const myolo_tensor* y, const float* saved,
                                  const float* gamma, const float* beta, int act, float* dsum,
                                  const myolo_bn_split* split, void* stream);
int myolo_bn_act_bwd_apply_split(const myolo_tensor* gout, const myolo_tensor* y, const float* saved,
                                 const float* gamma, const float* beta, int act, const float* dsum,
                                 float* dgamma, float* dbeta, const myolo_tensor* dy,
                                 const myolo_tensor* gres, int gres_accumulate, const myolo_bn_split* split, void* stream);

/* ---- pooling / resampling / glue -------------------------------------------------------------- */
/* SPP: three stride-1 max pools k=5,9,13, -inf padding (common.py:170).  idx (optional, u8 [3][n,h,w,c])
 * records the first-max window offset for backward. */
int myolo_spp_pool_fwd(const myolo_tensor* x, const myolo_tensor* o5, const myolo_tensor* o9,
                       const myolo_tensor* o13, uint8_t* idx, void* stream);
int myolo_spp_pool_bwd(const myolo_tensor* g5, const myolo_tensor* g9, const myolo_tensor* g13,
                       const uint8_t* idx, const myolo_tensor* gx, int accumulate, void* stream);
/* nn.Upsample(None, 2, 'nearest') (yaml:31,36) or plain copy (scale=1) into a (slice) view */
int myolo_copy_up_fwd(const myolo_tensor* x, const myolo_tensor* out, int scale, void* stream);
int myolo_copy_up_bwd(const myolo_tensor* gout, const myolo_tensor* gx, int scale, int accumulate, void* stream);
/* bilinear, align_corners=True (yolo.py:57..174, common.py:534-537, detect.py:191).
 * bwd scratch (optional): fp32[n*h*w*c of gx], ZERO on entry, left dirty -- lets the PyramidPooling case (gx <= 6x6, thousands of
 * outputs per input pixel) split the reduction over ~512 workgroups instead of one per input pixel. */
int myolo_bilinear_fwd(const myolo_tensor* x, const myolo_tensor* out, void* stream);
int myolo_bilinear_bwd(const myolo_tensor* gout, const myolo_tensor* gx, int accumulate, float* scratch, void* stream);
/* nn.AdaptiveAvgPool2d(k) (common.py:521-524, 214): out [n,k,k,c].  scratch (optional): fp32[n*k*k*c] ZEROED by the caller;
 * with it, big bins are summed by many workgroups in parallel. */
int myolo_adaptive_avgpool_fwd(const myolo_tensor* x, const myolo_tensor* out, float* scratch, void* stream);
/* PyramidPooling's pools (common.py:521-524: AdaptiveAvgPool2d(1), (2), (3), (6) of the SAME map) in one pass over x: outs[0..count)
 * (count <= 4; 1 = a single pool, e.g. FFM's global average) are the [n,k,k,c] results, scratch fp32 [8][n][sum of k*k][c] ZEROED by
 * the caller (left dirty; 8 replicas spread the same-address atomics).  MYOLO_EINVAL when the
 * map is too narrow for the kernel's bin bookkeeping (x.w / 256*seg/c + 2 > x.w / kmax): use the single-pool entry point then. */
int myolo_adaptive_avgpool_fwd_multi(const myolo_tensor* x, const myolo_tensor* outs, int count, float* scratch, void* stream);
int myolo_adaptive_avgpool_bwd(const myolo_tensor* gout, const myolo_tensor* gx, int accumulate, void* stream);
/* the backward of up to four adaptive average pools of ONE input (PyramidPooling, common.py:521-524): gouts = array of `count`
 * pooled-gradient views; gx (+)= sum over them -- the input gradient is read-modified-written once instead of once per pool */
/* PyramidPooling (common.py:534-537): the backward of `count` (<= 4) bilinear align_corners upsamples of k x k maps (k <= 6) whose
 * outputs are ADJACENT equal-width channel slices of one tensor (`gout`: the view over all of them, <= 128 channels in fp16): one
 * pass over gout, gxs[t] (+)= result.  scratch: fp32 [sum of the gxs element counts], ZERO on entry, left dirty. */
/* forward of the same: `count` bilinear align_corners upsamples of xs[t] into the adjacent equal-width channel slices of `out` */
int myolo_pyramid_upsample_fwd(const myolo_tensor* xs, int count, const myolo_tensor* out, void* stream);
int myolo_pyramid_upsample_bwd(const myolo_tensor* gout, const myolo_tensor* gxs, int count, const int32_t* accumulate, float* scratch,
                               void* stream);
int myolo_adaptive_avgpool_bwd_multi(const myolo_tensor* gouts, int count, const myolo_tensor* gx, int accumulate, void* stream);
/* ---- 1x1 Conv2d (+ train-mode BatchNorm2d) (+ activation) on a map of at most MYOLO_TINY_MAX_PIX pixels, forward and backward,
 * ONE workgroup per layer and up to MYOLO_TINY_MAX_GROUP independent layers per launch (csrc/tiny_conv.hip).  Replaces, for
 * PyramidPooling's four `Conv(in_channels, in_channels // 4, k=1)` on the 1x1 .. 6x6 pooled maps (models/common.py:521-537; Conv =
 * nn.Conv2d + nn.BatchNorm2d + nn.SiLU, common.py:34-46), myolo_conv + myolo_bn_act_fwd in the forward and myolo_bn_act_bwd_reduce +
 * myolo_bn_act_bwd_apply + the dgrad myolo_conv in the backward (autograd of the same); the weight gradient stays myolo_conv_wgrad over
 * the `dy` the backward writes.  w: the OIHW fp32 master weight [cout][cin] (cast to the plan dtype inside, = autocast).
 * gamma == NULL: no BatchNorm (z = conv, out = act(z)).  Constraints (else MYOLO_EINVAL): every tensor of one dtype, 16-byte aligned
 * views; pixels <= MYOLO_TINY_MAX_PIX; cout <= 128; fp16: cin % 32 == 0, cin <= 512, cout % 16 == 0; fp32: cin <= 512. */
#define MYOLO_TINY_MAX_PIX 1024
#define MYOLO_TINY_MAX_GROUP 4
typedef struct myolo_tiny_conv_desc {
  myolo_tensor x;            /* [n,h,w,cin] */
  myolo_tensor z;            /* [n,h,w,cout] raw conv output: written by the forward, read by the backward */
  myolo_tensor out;          /* forward: act(bn(z)) */
  const float* w;
  const float* gamma;        /* BatchNorm weight / bias (fp32[cout]) or NULL */
  const float* beta;
  float*   running_mean;     /* updated by the forward like myolo_bn_act_fwd (may be NULL) */
  float*   running_var;
  int64_t* nbt;              /* num_batches_tracked += 1 (may be NULL) */
  float*   saved;            /* fp32[2*cout]: batch mean, invstd -- forward writes, backward reads */
  float    eps, momentum;
  int32_t  act;
  int32_t  gx_accumulate;    /* backward: 1 = gx += result */
  myolo_tensor gout;         /* backward: gradient w.r.t. out */
  myolo_tensor dy;           /* backward OUT: gradient w.r.t. z (the weight gradient's operand) */
  myolo_tensor gx;           /* backward OUT: gradient w.r.t. x; ptr == NULL: not needed */
  float*   dgamma;           /* += (may be NULL) */
  float*   dbeta;
} myolo_tiny_conv_desc;
/* 1 when a layer of this shape fits the kernels (pixel / channel limits and the workgroup's LDS, forward and backward), else 0: what a
 * planner asks before it takes a layer off the myolo_conv / myolo_bn_act_* chain (engine.ConvOp.tiny_ok) */
int myolo_tiny_conv_ok(int dtype, int pixels, int cin, int cout);
int myolo_tiny_conv_fwd(const myolo_tiny_conv_desc* d, int n, void* stream);
int myolo_tiny_conv_bwd(const myolo_tiny_conv_desc* d, int n, void* stream);
/* FFM gate: out = feat*att + feat, att [n,1,1,c] (common.py:228-229) */
int myolo_gate_fwd(const myolo_tensor* feat, const myolo_tensor* att, const myolo_tensor* out, void* stream);
int myolo_gate_bwd(const myolo_tensor* gout, const myolo_tensor* feat, const myolo_tensor* att,
                   const myolo_tensor* gfeat, int accumulate, float* gatt_f32 /* [n*c] zeroed by caller */,
                   void* stream);
/* ARM / Attention gate: out = feat*att (torch.mul(feat, atten), common.py:192,207) -- the FFM gate without the `+ feat` */
int myolo_gate_mul_fwd(const myolo_tensor* feat, const myolo_tensor* att, const myolo_tensor* out, void* stream);
int myolo_gate_mul_bwd(const myolo_tensor* gout, const myolo_tensor* feat, const myolo_tensor* att,
                       const myolo_tensor* gfeat, int accumulate, float* gatt_f32 /* [n*c] zeroed by caller */,
                       void* stream);
/* out (+)= a  (elementwise on views; BiSe `m16 + feat3`, gradient fan-in) */
int myolo_add(const myolo_tensor* a, const myolo_tensor* out, int accumulate, void* stream);
int myolo_fill_zero(const myolo_tensor* t, void* stream);
/* fp32 [n*c] -> view (cast), used for small fp32 side results */
int myolo_cast_from_f32(const float* src, const myolo_tensor* out, void* stream);

/* train-mode nn.Dropout (yolo.py:65,140): keep-mask u8 per element (dense [n,h,w,c]); `counter` is a 64-bit draw
 * counter in device memory, advanced by the call itself (graph-replay safe). */
int myolo_dropout_fwd(const myolo_tensor* x, const myolo_tensor* out, uint8_t* mask, float p, uint64_t* counter,
                      void* stream);
int myolo_dropout_bwd(const myolo_tensor* gout, const uint8_t* mask, const myolo_tensor* gx, float p, int accumulate,
                      void* stream);

/* ---- head outputs ------------------------------------------------------------------------------ */
/* final nn.Upsample(x8, bilinear, align_corners=True) of the class logits (yolo.py:67,118,143,163) into a
 * [N,C,H,W]-logical tensor with arbitrary element strides (sn,sc,sh,sw); bwd is its transpose.
 * scale (optional, device float[1]): the incoming gradient is g * scale[0] (see myolo_seg_ce_fwd_grad). */
int myolo_seg_upsample_fwd(const myolo_tensor* low, void* out, int out_dtype, int H, int W,
                           int64_t sn, int64_t sc, int64_t sh, int64_t sw, void* stream);
int myolo_seg_upsample_bwd(const void* g, int g_dtype, int H, int W, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                           const myolo_tensor* glow, int accumulate, const float* scale, void* stream);
/* detect.py:191-193 fused: bilinear resize of the logits to (H,W) + argmax over classes -> labels [N,H,W] (u8|i64) */
int myolo_seg_argmax(const myolo_tensor* low, void* labels, int label_dtype, int H, int W, void* stream);
/* seg_validation counters (utils/metrics.py:234-275 batch_pix_accuracy + batch_intersection_union, test.py:31-65) from predicted
 * labels (u8|i64, e.g. myolo_seg_argmax) and int64 targets (-1 = ignore): counts (device uint64[2+3*nclass], zeroed here) =
 * {pixel_correct, pixel_labeled, area_inter[nclass], area_pred[nclass], area_lab[nclass]}; union = pred + lab - inter. */
int myolo_seg_metrics(const void* pred, int pred_dtype, const int64_t* target, int64_t total, int nclass, uint64_t* counts,
                      void* stream);
/* gradient of Detect's view/permute (yolo.py:214): dense [N,na,ny,nx,no] -> NHWC view [N,ny,nx,>=na*no] */
int myolo_detect_unpermute(const void* g, int g_dtype, int na, int no, const myolo_tensor* out, void* stream);
/* Detect eval decode (yolo.py:216-223) of one level: raw dense [N,na,ny,nx,no] -> rows [row0, row0+na*ny*nx) of
 * z dense [N,a_total,no]; anchor_wh_px = anchor_grid[i] (pixels), host pointer, na*2 floats */
int myolo_detect_decode(const void* raw, int dtype, int n, int na, int ny, int nx, int no, float stride,
                        const float* anchor_wh_px, void* z, int64_t a_total, int64_t row0, void* stream);

/* ---- losses ------------------------------------------------------------------------------------ */
/* nn.CrossEntropyLoss(ignore_index) over [N,C,H,W]-logical logits with element strides (utils/loss.py:236-237).
 * acc (double[2], device): zeroed here, then acc[0] = sum of per-pixel losses, acc[1] = number of valid pixels.
 * pix (optional, float[N*H*W]): per-pixel loss as reduction='none' gives (0 at ignored pixels) -- OhemCELoss input.
 * loss (optional, float[1]): acc[0]/acc[1]. */
int myolo_seg_ce_fwd(const void* logits, int dtype, int n, int c, int h, int w, int64_t sn, int64_t sc, int64_t sh,
                     int64_t sw, const int64_t* target, int ignore_index, double* acc, float* pix, float* loss,
                     void* stream);
/* myolo_seg_ce_fwd fused with the backward of the plain mean-CE case, for dense channels-last logits ([N,H,W,C] storage, 16-byte
 * aligned; anything else is MYOLO_EINVAL): additionally writes grad (same layout/dtype) = softmax - onehot, 0 on ignored
 * pixels, i.e. d(loss)/d(logits) up to the scalar gout/acc[1], which exists only after the reduction over all pixels.
 * myolo_seg_ce_scale computes that scalar on device (scale[0] = gout[0]/acc[1]) for the gradient's consumer
 * (myolo_seg_upsample_bwd `scale`): the logits are read once per step instead of twice and no rescaling pass is needed. */
int myolo_seg_ce_fwd_grad(const void* logits, void* grad, int dtype, int n, int c, int h, int w, const int64_t* target,
                          int ignore_index, double* acc, float* loss, void* stream);
int myolo_seg_ce_scale(const double* acc, const float* gout, float* scale, void* stream);
/* The head's final nn.Upsample (bilinear, align_corners=True, yolo.py:163) + the mean cross entropy + its gradient + the transposed
 * upsample in ONE pass over the LOW-resolution class logits `low` ([n,h,w,19] view): the [N,19,H,W] logits are recomputed per pixel
 * in registers (rounded to `low`'s dtype, like the tensor the reference's loss reads) and neither they nor their gradient touch HBM.
 * acc / loss as myolo_seg_ce_fwd; glow32: dense fp32 [n,h,w,19], zeroed here, receives d(sum of pixel losses)/d(low) -- the
 * classifier's gradient up to the scalar gout/acc[1] (myolo_seg_ce_scale), applied by myolo_seg_lowgrad_apply:
 *   glow (+)= glow32 * scale[0]   (glow: the [n,h,w,19] gradient view; scale optional).
 * MYOLO_EINVAL for a class count other than 19 (callers take myolo_seg_upsample_fwd + myolo_seg_ce_fwd_grad instead). */
int myolo_seg_upce_fwd_grad(const myolo_tensor* low, int H, int W, const int64_t* target, int ignore_index, double* acc,
                            float* loss, float* glow32, void* stream);
int myolo_seg_lowgrad_apply(const float* glow32, const myolo_tensor* glow, int accumulate, const float* scale, void* stream);
/* OhemCELoss (utils/loss.py:303-328) over the LOW-resolution logits of the head, the x8 bilinear upsample (yolo.py:163) folded in like
 * myolo_seg_upce_fwd_grad: _pix writes the per-pixel losses (float [n][H][W], 0 on ignored pixels) and acc[1] = valid count;
 * myolo_ohem_select then picks the hard pixels; _grad recomputes every pixel's softmax and folds (softmax - onehot) * selection weight
 * into glow32 (fp32 [n][h][w][19], overwritten) -- the caller scales by gout / sel.denom (myolo_seg_lowgrad_apply).  19 classes. */
int myolo_seg_upce_ohem_pix(const myolo_tensor* low, int H, int W, const int64_t* target, int ignore_index, double* acc, float* pix,
                            void* stream);
int myolo_seg_upce_ohem_grad(const myolo_tensor* low, int H, int W, const int64_t* target, int ignore_index, const float* pix,
                             const float* sel, float thresh, float* glow32, void* stream);
/* OhemCELoss.forward_once (utils/loss.py:321-328) on the per-pixel losses: mean of losses > thresh, or, if fewer than
 * n_min = acc[1]//16 qualify, mean of the n_min largest (device radix select, no host sync).
 * st: double[5] scratch, ws: uint32[2052] scratch, loss: float[1], sel: float[4] selection record for the backward. */
int myolo_ohem_select(const float* pix, int64_t total, float thresh, const double* acc, double* st, uint32_t* ws,
                      float* loss, float* sel, void* stream);
/* d loss / d logits = (softmax - onehot) * gout[0] / denom on the selected pixels (all valid pixels when sel == NULL). */
int myolo_seg_ce_bwd(const void* logits, void* grad, int dtype, int n, int c, int h, int w, int64_t sn, int64_t sc,
                     int64_t sh, int64_t sw, int64_t gsn, int64_t gsc, int64_t gsh, int64_t gsw, const int64_t* target,
                     int ignore_index, const double* acc, const float* gout, const float* pix, const float* sel,
                     float thresh, void* stream);

/* ComputeLoss.__call__ + build_targets + bbox_iou(CIoU) (utils/loss.py:115-217, utils/general.py:343-380). */
typedef struct myolo_detloss_desc {
  int32_t nl, na, no, bs, nt, dtype;       /* levels, anchors/level, 5+nc, batch, target rows, dtype of p / gp */
  const void* p[5];                        /* Detect training outputs [bs,na,ny,nx,no], dense */
  void*       gp[5];                       /* their gradients (bwd only) */
  int32_t ny[5], nx[5];
  const float* anchors;                    /* [nl,na,2] in grid units (Detect.anchors, yolo.py:262), device */
  const float* targets;                    /* [nt,6] (image, class, x, y, w, h) normalised, device */
  float balance[5];                        /* loss.py:109 */
  float box, obj, cls, cls_pw, obj_pw, anchor_t, gr, cp, cn;
  int32_t* winner;                         /* workspace: int32 per cell of all levels */
  float*   ciou;                           /* workspace: nl * 5*na*nt floats */
  double*  acc;                            /* workspace: double[20] */
  float*   out;                            /* float[5]: loss*bs, lbox, lobj, lcls, loss (loss.py:156-162) */
  float*   gp32;                           /* bwd workspace when dtype is F16: fp32 copy of all gradients (atomics) */
  const float* gout;                       /* bwd: d(objective)/d(out[0]), device scalar */
} myolo_detloss_desc;
int myolo_detloss_fwd(const myolo_detloss_desc* d, void* stream);
int myolo_detloss_bwd(const myolo_detloss_desc* d, void* stream);

/* ---- optimizer side (multi-tensor, one launch for all tensors) ---------------------------------- */
/* table: device int64 [ntensors][6] = {ptr0, ptr1, ptr2, numel, group, 0} (fp32 tensors); chunks: device int32 [nchunks][2]
 * = {tensor index, first element}; workgroup b covers chunk_elems elements of its tensor. */
typedef struct myolo_sgd_hyper {
  float lr[8], momentum[8], weight_decay[8];    /* per param group (train.py:121-137: BN weights | weights + decay | biases) */
  int32_t nesterov;
  int32_t reserved;
} myolo_sgd_hyper;
/* torch.optim.SGD step (train.py:397): ptr0 = param, ptr1 = grad, ptr2 = momentum buffer (zero-initialised).
 * scale (optional, device float): gradients are divided by scale[0] (AMP unscale);
 * found_inf (optional, device float): the whole update is skipped when found_inf[0] != 0 (GradScaler.step). */
int myolo_mt_sgd(const int64_t* table, const int32_t* chunks, int nchunks, int chunk_elems, const myolo_sgd_hyper* hyper,
                 const float* scale, const float* found_inf, void* stream);
/* found_inf[0] = 1 if any element of ptr<which> is inf/nan (never cleared here). */
int myolo_mt_check_finite(const int64_t* table, const int32_t* chunks, int nchunks, int chunk_elems, int which,
                          float* found_inf, void* stream);
/* ModelEMA.update (utils/torch_utils.py:290-300): ptr0 = ema tensor, ptr1 = model tensor: v = v*decay + (1-decay)*m */
int myolo_mt_ema(const int64_t* table, const int32_t* chunks, int nchunks, int chunk_elems, float decay, void* stream);
/* GradScaler.update: backoff on found_inf, growth after `interval` clean steps; clears found_inf. */
int myolo_scaler_update(float* scale, int32_t* growth_tracker, float* found_inf, float growth, float backoff, int interval,
                        void* stream);

/* ---- inference post-processing ------------------------------------------------------------------- */
/* non_max_suppression (utils/general.py:421-509) + torchvision.ops.nms (call site general.py:493), all images at once.
 * pred: dense [batch, A, no] (xywh, obj, cls...), f16|f32 (arithmetic is fp32).  Workspaces (device, caller-owned):
 * counts int32[batch], cand float[batch][cap][6], cand_idx int32[batch][cap], sorted float[batch][max_nms][6];
 * cap >= A (A*nc when multi_label).  Results: out float[batch][max_det][6] = (x1,y1,x2,y2,conf,cls) in descending conf,
 * nkeep int32[batch].  class_mask: the `classes=` filter (general.py:476-477) as a bit set, bit j = keep class j; 0 = keep all
 * (nc <= 64 when non-zero).  sort_ws (optional, device int32 [batch][3*65536 + cap]): the descending-score order is produced by a
 * counting sort instead of the O(n^2) rank kernel -- for test.py's conf 0.001 / multi_label lists (1e5 candidates per image).
 * mws (optional, device, 16-byte aligned, >= myolo_nms_ws_bytes(batch, cap) bytes; single-label calls, cap == A, max_det <= 320):
 * images with at most 8192 candidates (detect.py) take the device-wide path -- rank, scatter and the suppression bit matrix on all
 * CUs, class by class (classes cannot suppress each other once box + cls*max_wh is formed, as long as the un-offset coordinates span
 * less than max_wh: checked per image), one wave per (image, class) walks its part of the sorted list with bit operations only, the
 * kept boxes are merged back into score order; longer lists, images failing that check and callers without mws keep the
 * single-workgroup lazy scan.  Identical results either way. */
int64_t myolo_nms_ws_bytes(int batch, int cap);
int myolo_nms(const void* pred, int dtype, int batch, int A, int no, float conf_thres, float iou_thres, int multi_label,
              int agnostic, float max_wh, int max_nms, int max_det, int cap, int32_t* counts, float* cand,
              int32_t* cand_idx, float* sorted, float* out, int32_t* nkeep, uint64_t class_mask, int32_t* sort_ws, void* mws,
              int64_t mws_bytes, void* stream);

/* test.py:230-262 (true-positive matrix of one image, the input of ap_per_class): pred [n][6] = (x1,y1,x2,y2,conf,cls) in NMS order and
 * labels [m][5] = (cls,x1,y1,x2,y2), native image space, device float32; iouv device float[niou] (test.py:98 linspace(0.5,0.95,10));
 * correct device uint8 [n][niou].  ws: device scratch, 8-byte aligned, >= 8*n + 16*ceil(m/16) bytes.  box_iou arithmetic of
 * utils/general.py:388-410 in its evaluation order (fp32, IEEE division); no host synchronisation. */
int myolo_match_predictions(const float* pred, int n, const float* labels, int m, const float* iouv, int niou, uint8_t* correct,
                            void* ws, int64_t ws_bytes, void* stream);

/* ---- detect.py frame pipeline ------------------------------------------------- */
/* uint8 HWC frame (h0 x w0 x 3) -> model input [1,3,H,W] NCHW f16|f32: constant border placement at (top,left) = the padding half of
 * `letterbox` (utils/datasets.py:840-847, no resampling), channel reversal `img[:, :, ::-1].transpose(2, 0, 1)` (datasets.py:185) when
 * swap_rb, and detect.py:136-137's uint8 -> dtype -> /255 through lut256 (device, 256 values of out_dtype, supplied by the caller
 * as torch computes them: bit-identical normalisation). */
int myolo_frame_pack(const uint8_t* frame_hwc, int h0, int w0, int swap_rb, int top, int left, int H, int W, int pad_value,
                     void* out_nchw, int out_dtype, const void* lut256, void* stream);
/* the same with the frame resampled to (rh, rw) first -- letterbox's cv2.resize(img, new_unpad, interpolation=cv2.INTER_LINEAR)
 * (datasets.py:843-844) on 8-bit images: OpenCV's 1/2048 fixed-point bilinear (exact 2x down-scale = 2x2 box average), then border,
 * channel swap, HWC->CHW and the /255 table as in myolo_frame_pack */
int myolo_frame_resize_pack(const uint8_t* frame_hwc, int h0, int w0, int rh, int rw, int swap_rb, int top, int left, int H, int W,
                            int pad_value, void* out_nchw, int out_dtype, const void* lut256, void* stream);
/* detect.py:193-194: mask = colormap[label] (channel-reversed when swap_rb = `label2image(...)[:, :, ::-1]`), dst =
 * cv2.addWeighted(mask, alpha, im0, beta, gamma) on uint8 (float32 products and sums rounded separately, round-half-even,
 * saturated).  labels u8|i64 [h,w] (clamped to [0,ncls)); colormap_rgb device uint8[ncls][3]; mask_hwc / dst_hwc: either may be NULL. */
int myolo_seg_blend(const void* labels, int label_dtype, const uint8_t* im0_hwc, int h, int w, const uint8_t* colormap_rgb, int ncls,
                    int swap_rb, float alpha, float beta, float gamma, uint8_t* mask_hwc, uint8_t* dst_hwc, void* stream);

/* ---- training-time augmentation on the device ------------------------------------------------------------ */
/* SegmentationDataset.py:118-151 `_sync_transform` for one sample, fused (only the crop is computed): img.transpose(FLIP_LEFT_RIGHT)
 * when flip; img.resize((ow, oh), BILINEAR) / mask.resize((ow, oh), NEAREST); ImageOps.expand to the crop size (fill 0 / 255);
 * crop (x1, y1, x1+wc, y1+hc); `_mask_transform` (label id -> train id through lab_lut[256], int64).
 * Pillow's 8-bit resampler: the caller supplies its per-output-pixel tables exactly as precompute_coeffs() / normalize_coeffs_8bpc()
 * (Resample.c) build them -- hb/vb: int32 [ow|oh][2] = (first source index, tap count); hk/vk: int32 [ow|oh][ksh|ksv] coefficients in
 * 2^-22 units (identity tables when a dimension is not resized) -- and ImagingScaleAffine's NEAREST index tables xin [ow], yin [oh].
 * img: uint8 [H0][W0][3]; mask: uint8 [H0][W0]; out_img: uint8 [hc][wc][3] (or NULL); out_lab: int64 [hc][wc] (or NULL).
 * All pointers are device memory. */
typedef struct myolo_seg_sync_desc {
  const uint8_t* img;
  const uint8_t* mask;
  int32_t H0, W0, flip, ow, oh;
  int32_t ksh, ksv;
  int32_t x1, y1, wc, hc;
  int32_t reserved;
  const int32_t* hb;
  const int32_t* hk;
  const int32_t* vb;
  const int32_t* vk;
  const int32_t* xin;
  const int32_t* yin;
  uint8_t* out_img;
  int64_t* out_lab;
  const int64_t* lab_lut;
} myolo_seg_sync_desc;
int myolo_seg_sync_transform(const myolo_seg_sync_desc* d, void* stream);
/* torchvision.transforms.ColorJitter on a PIL RGB image + ToTensor (get_citys_loader, SegmentationDataset.py:462-466): the four
 * adjustments in the order order4[0..3] (0 brightness, 1 contrast, 2 saturation, 3 hue, -1 = skip), Pillow's arithmetic restated
 * (ImageEnhance / Blend.c float32 blend with (UINT8) truncation, 'L' = (19595R + 38470G + 7471B + 0x8000) >> 16, contrast grey level
 * int(mean(L) + 0.5) of the image AT THAT STAGE, Convert.c rgb2hsv / hsv2rgb); hue_shift_u8 = the uint8 torchvision adds to the H plane.
 * img_hwc uint8 [h][w][3]; scratch: device uint64[1] (needed when contrast is in the order); out_hwc: uint8 [h][w][3] or NULL;
 * out_chw: [3][h][w] of out_dtype = lut256[value] (the caller's 256 values of v/255: ToTensor) or NULL. */
int myolo_color_jitter(const uint8_t* img_hwc, int h, int w, const int32_t* order4, float brightness, float contrast, float saturation,
                       int hue_shift_u8, uint64_t* scratch, uint8_t* out_hwc, void* out_chw, int out_dtype, const void* lut256,
                       void* stream);

/* cv2.resize(img, (rw, rh), interpolation=cv2.INTER_LINEAR) on a uint8 HWC image (load_image, utils/datasets.py:638-640), the
 * arithmetic of myolo_frame_resize_pack with HWC uint8 output */
int myolo_resize_u8(const uint8_t* img_hwc, int h0, int w0, int rh, int rw, uint8_t* out_hwc, void* stream);
/* One detection training sample, fused (utils/datasets.py): load_mosaic's canvas (672-711: up to 4 source images pasted at
 * [y1a:y2a, x1a:x2a] = img[y1a-padh : , x1a-padw : ] on a cw x ch canvas of `fill`) -> random_perspective's cv2.warpAffine(img, M[:2],
 * dsize=(ow, oh), borderValue=fill) (851-895; M = the dst->src matrix, i.e. ALREADY inverted the way cv::warpAffine inverts its
 * argument; warp = 0 copies the canvas) -> augment_hsv (646-658; hsv_lut = uint8 [3][256] hue/sat/val tables, NULL = skip) -> flipud /
 * fliplr (574-584) -> out_chw uint8 [3][oh][ow] RGB (590: img[:, :, ::-1].transpose(2, 0, 1)) and/or out_hwc uint8 [oh][ow][3] BGR.
 * A single source covering the canvas gives the non-mosaic path (letterboxed image -> random_perspective). */
typedef struct myolo_mosaic_src {
  const uint8_t* img;            /* uint8 [h][w][3] BGR, device */
  int32_t h, w;
  int32_t x1a, y1a, x2a, y2a;    /* window on the canvas */
  int32_t padw, padh;            /* canvas = source + pad */
} myolo_mosaic_src;
typedef struct myolo_mosaic_desc {
  myolo_mosaic_src src[4];
  int32_t nsrc, cw, ch, warp;
  double  M[6];
  int32_t ow, oh, fliplr, flipud, fill, reserved;
  const uint8_t* hsv_lut;
  uint8_t* out_chw;
  uint8_t* out_hwc;
} myolo_mosaic_desc;
int myolo_mosaic_warp(const myolo_mosaic_desc* d, void* stream);

/* ---- native plan executor (csrc/plan_exec.hip) ---------------------------------------------------------
 * Replaces the reference's Python interpreter loop over the layers (models/yolo.py:293-316 forward_once: one module call per
 * layer, each several ATen launches) AND this package's own round-1/2 loop of one ctypes call per launch: a plan's forward or
 * backward launch list is serialised once into myolo_prog_op records and one myolo_prog_run() call issues a [first, last) range
 * of them.  Every argument of the target entry point except its trailing stream occupies one 8-byte slot: pointers as addresses
 * (descriptor structs stay owned by the caller and must outlive the program), integers sign-extended, floats as their IEEE bit
 * pattern in the low 32 bits.  cond != 0: the op only runs while *(const int32_t*)cond == cond_val (host memory, read at run time;
 * the per-step choice between the full-resolution and the fused low-resolution segmentation gradient).
 * CALL_SIDE forks to `side_stream` behind an event recorded on `main_stream` (weight gradients: nothing on the backward chain
 * depends on them); JOIN makes main_stream wait for side_stream; MEMSET zeroes a[1] bytes at a[0] on main_stream.  With a NULL
 * side stream everything runs on main_stream.  No allocation, no synchronisation, no host read besides `cond`. */
#define MYOLO_PROG_MAX_ARGS 24
enum { MYOLO_OP_CALL = 0, MYOLO_OP_CALL_SIDE = 1, MYOLO_OP_JOIN = 2, MYOLO_OP_MEMSET = 3 };
typedef struct myolo_prog_op {
  int32_t  kind;                 /* MYOLO_OP_* */
  int32_t  fn;                   /* myolo_prog_fn_id() of the entry point (CALL / CALL_SIDE) */
  int32_t  nargs;                /* must equal myolo_prog_fn_nargs(fn) */
  int32_t  cond_val;
  uint64_t cond;                 /* host address of an int32 or 0 */
  uint64_t a[MYOLO_PROG_MAX_ARGS];
} myolo_prog_op;
int   myolo_prog_fn_id(const char* name);          /* -1: this entry point cannot be part of a program */
int   myolo_prog_fn_nargs(int fn);                 /* argument slots (the stream excluded) */
void* myolo_prog_create(const myolo_prog_op* ops, int n);      /* copies the records, creates the fork/join events; NULL on a bad record */
void  myolo_prog_destroy(void* prog);
uint64_t* myolo_prog_slot(void* prog, int op, int arg);        /* address of one argument slot (to re-bind caller tensors per run) */
int   myolo_prog_run(void* prog, int first, int last, void* main_stream, void* side_stream);  /* 0 or the failing launch's error */
int   myolo_prog_last_op(void* prog);              /* index of the op the last failing run stopped at */

/* ---- order between two HIP streams without an event (csrc/plan_exec.hip) --------------------------------------------------
 * The eval forward of models/yolo.py:293-316 has two independent tails behind the neck (Detect, the segmentation head): they run
 * on two streams.  A hipStreamWaitEvent between them costs 90-170 us per frame on this runtime; a counting semaphore in device
 * memory costs about one.  sem: MYOLO_QUEUE_SEM_BYTES of zeroed device memory, 4-byte aligned (word 0 = count, word 32 = polls
 * that ran into timeout_ms: the consumer went on WITHOUT its dependency -- the caller must check it before trusting results).
 * myolo_queue_post: a one-lane kernel on `stream` that adds 1 behind everything enqueued on it so far.  myolo_queue_wait: a
 * one-lane kernel on `stream` that polls until the count is positive, takes 1, and only then lets that stream's later kernels
 * start.  Both are capturable into hipGraphs.  Enqueue the post BEFORE the wait when both streams may share a hardware queue
 * (HIP maps streams onto GPU_MAX_HW_QUEUES queues): a poll in front of its own producer only ends by its timeout. */
#define MYOLO_QUEUE_SEM_BYTES 256
int myolo_queue_post(void* sem, void* stream);
int myolo_queue_wait(void* sem, int timeout_ms, void* stream);

/* ---- launch trace (test infrastructure; no reference counterpart) ----------------------------------------------------------
 * The dispatchers behind myolo_conv / myolo_conv_wgrad / ... pick a kernel family and a template variant (tile shape, ring
 * depth, epilogue flags) from the descriptor.  myolo_trace_start(1) clears the table and makes every launch site record
 * itself; myolo_trace_read copies "count<TAB>site<NEWLINE>" lines (site = the launcher's signature with its template
 * arguments) into buf and returns the bytes needed (buf may be NULL).  tests/test_gpu_bench_plan.py uses it to list which
 * variants the benchmarked plans run, next to a per-launch comparison of those launches with torch fp32. */
int     myolo_trace_start(int on);
int64_t myolo_trace_read(char* buf, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* MYOLO_H_ */
