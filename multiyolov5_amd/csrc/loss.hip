// Losses of the joint det+seg step, fused per head (all HBM-bound streaming kernels, fp32 math, double accumulators).
//   seg_ce_fwd / seg_ce_bwd : nn.CrossEntropyLoss(ignore_index) over [N,C,H,W] logits (reference utils/loss.py:236-237)
//                             and OhemCELoss (utils/loss.py:303-328: per-pixel CE, keep loss > -log(thresh), else the
//                             n_min = valid//16 hardest) -- the top-k fallback is a 3-pass radix select on device.
//   detloss_fwd / detloss_bwd : ComputeLoss.__call__ + build_targets + bbox_iou(CIoU) (utils/loss.py:115-217,
//                             utils/general.py:343-380): anchor matching, 5-neighbour expansion, CIoU, objectness target
//                             scatter with the CPU's "last write wins" order, BCE obj / cls, and all gradients -- no host
//                             sync, no boolean-mask indexing.
#include "myolo_dev.h"
#include <map>
#include <mutex>
#include <utility>

namespace {

struct Strided4 { void* ptr; int64_t sn, sc, sh, sw; int dtype; };
__device__ __forceinline__ float ld_any(const void* p, int64_t i, int dt) {
  return dt == MYOLO_F16 ? (float)((const half_t*)p)[i] : ((const float*)p)[i];
}
__device__ __forceinline__ void st_any(void* p, int64_t i, int dt, float v) {
  if (dt == MYOLO_F16) ((half_t*)p)[i] = (half_t)v; else ((float*)p)[i] = v;
}

constexpr int MAXC = 32;

__device__ __forceinline__ double block_sum256(double v, double* sh) {
  // wave reduce then across the 4 waves
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) r = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return r;
}

// ------------------------------------------------------------------------------------------ segmentation CE
// acc[0] += sum of per-pixel losses over valid pixels, acc[1] += number of valid pixels.
// pix (optional): per-pixel loss (0 at ignored pixels), as CrossEntropyLoss(reduction='none') gives (loss.py:311,323)
__global__ __launch_bounds__(256) void seg_ce_fwd_kernel(Strided4 x, const int64_t* tgt, int C, int H, int W, int64_t total,
                                                         int ignore, double* acc, float* pix) {
  __shared__ double sh[4];
  double lsum = 0.0, lcnt = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W); const int y = (int)((i / W) % H); const int n = (int)(i / ((int64_t)W * H));
    const int64_t t = tgt[i];
    float l = 0.f;
    if (t != ignore) {
      const int64_t b = (int64_t)n * x.sn + (int64_t)y * x.sh + (int64_t)xx * x.sw;
      float v[MAXC];
      float m = -INFINITY;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) { v[c] = ld_any(x.ptr, b + (int64_t)c * x.sc, x.dtype); m = fmaxf(m, v[c]); }
      float s = 0.f, xt = 0.f;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) { s += expf(v[c] - m); if (c == (int)t) xt = v[c]; }
      l = (m + logf(s)) - xt;
      lsum += (double)l;
      lcnt += 1.0;
    }
    if (pix) pix[i] = l;
  }
  const double bs = block_sum256(lsum, sh);
  const double bc = block_sum256(lcnt, sh);
  if (threadIdx.x == 0) { atomicAdd(acc + 0, bs); atomicAdd(acc + 1, bc); }
}

// dlogit = (softmax - onehot) * w,  w = gout / denom for the selected pixels:
//   plain CE : every valid pixel, denom = acc[1]
//   OHEM     : sel[0] = mode (0: loss > thresh, 1: top-k), denom = sel[1];  top-k: loss > kth -> 1, loss == kth -> tie weight sel[3]
struct OhemSel { float mode, denom, kth, tie_w; };
__global__ __launch_bounds__(256) void seg_ce_bwd_kernel(Strided4 x, Strided4 g, const int64_t* tgt, int C, int H, int W,
                                                         int64_t total, int ignore, const double* acc, const float* gout,
                                                         const float* pix, const OhemSel* sel, float thresh) {
  const float go = gout[0];
  float wbase;
  int mode = -1;
  float kth = 0.f, tie_w = 0.f;
  if (sel) {
    mode = (int)sel->mode; kth = sel->kth; tie_w = sel->tie_w;
    wbase = go / sel->denom;
  } else {
    wbase = (float)((double)go / acc[1]);
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W); const int y = (int)((i / W) % H); const int n = (int)(i / ((int64_t)W * H));
    const int64_t t = tgt[i];
    const int64_t gb = (int64_t)n * g.sn + (int64_t)y * g.sh + (int64_t)xx * g.sw;
    float w = (t != ignore) ? wbase : 0.f;
    if (mode == 0) { if (!(pix[i] > thresh)) w = 0.f; }
    else if (mode == 1) { const float l = pix[i]; w = l > kth ? wbase : (l == kth ? wbase * tie_w : 0.f); if (t == ignore) w = 0.f; }
    if (w == 0.f) {
#pragma unroll
      for (int c = 0; c < MAXC; ++c) if (c < C) st_any(g.ptr, gb + (int64_t)c * g.sc, g.dtype, 0.f);
      continue;
    }
    const int64_t b = (int64_t)n * x.sn + (int64_t)y * x.sh + (int64_t)xx * x.sw;
    float v[MAXC];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) { v[c] = ld_any(x.ptr, b + (int64_t)c * x.sc, x.dtype); m = fmaxf(m, v[c]); }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) if (c < C) { v[c] = expf(v[c] - m); s += v[c]; }
    const float inv = 1.f / s;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) st_any(g.ptr, gb + (int64_t)c * g.sc, g.dtype, (v[c] * inv - (c == (int)t ? 1.f : 0.f)) * w);
  }
}

// ---- OHEM selection (loss.py:321-328) -------------------------------------------------------------------
// ws layout (uint32 words): [0..2048) histogram, [2048] prefix bits, [2049] remaining k, [2050] pass shift
// state (double): st[0] = sum(loss > thresh), st[1] = count(loss > thresh), st[2] = sum(loss > kth), st[3] = count(loss > kth),
//                 st[4] = count(loss == kth)
__global__ __launch_bounds__(256) void ohem_thresh_kernel(const float* pix, int64_t total, float thresh, double* st) {
  __shared__ double sh[4];
  double s = 0.0, c = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float l = pix[i];
    if (l > thresh) { s += (double)l; c += 1.0; }
  }
  const double bs = block_sum256(s, sh), bc = block_sum256(c, sh);
  if (threadIdx.x == 0) { atomicAdd(st + 0, bs); atomicAdd(st + 1, bc); }
}
// radix select of the k-th largest loss (losses are >= 0, so their IEEE bit patterns order like unsigned ints):
// pass p histograms bits [shift, shift+nb) of the keys whose higher bits equal `prefix`
__global__ __launch_bounds__(256) void ohem_hist_kernel(const float* pix, int64_t total, uint32_t* ws, int shift, int nb,
                                                        uint32_t himask) {
  __shared__ uint32_t h[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) h[i] = 0;
  __syncthreads();
  const uint32_t prefix = ws[2048];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t k = __float_as_uint(pix[i]);
    if ((k & himask) == prefix) atomicAdd(&h[(k >> shift) & ((1u << nb) - 1)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 256) if (h[i]) atomicAdd(&ws[i], h[i]);
}
// one workgroup: walk the histogram from the top bin down until the remaining k falls inside a bin
__global__ __launch_bounds__(256) void ohem_pick_kernel(uint32_t* ws, int shift, int nb, const double* acc, int first) {
  __shared__ uint32_t h[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) h[i] = ws[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t k = first ? (uint32_t)((int64_t)acc[1] / 16) : ws[2049];   // n_min = valid // 16 (loss.py:322)
    if (k == 0) k = 1;                                                   // empty top-k: keep the select well-defined
    const int bins = 1 << nb;
    int b = bins - 1;
    for (; b > 0; --b) {
      if (h[b] >= k) break;
      k -= h[b];
    }
    ws[2048] |= ((uint32_t)b) << shift;
    ws[2049] = k;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 256) ws[i] = 0;
}
__global__ __launch_bounds__(256) void ohem_topk_sum_kernel(const float* pix, int64_t total, const uint32_t* ws, double* st) {
  __shared__ double sh[4];
  const float kth = __uint_as_float(ws[2048]);
  double s = 0.0, c = 0.0, e = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float l = pix[i];
    if (l > kth) { s += (double)l; c += 1.0; }
    else if (l == kth) e += 1.0;
  }
  const double bs = block_sum256(s, sh), bc = block_sum256(c, sh), be = block_sum256(e, sh);
  if (threadIdx.x == 0) { atomicAdd(st + 2, bs); atomicAdd(st + 3, bc); atomicAdd(st + 4, be); }
}
// loss value + selection record for backward
__global__ void ohem_final_kernel(const double* acc, const double* st, const uint32_t* ws, float* loss, OhemSel* sel) {
  const double n_min = (double)((int64_t)acc[1] / 16);
  if (st[1] >= n_min) {                      // enough hard pixels above the threshold (loss.py:325-326 not taken)
    loss[0] = (float)(st[0] / st[1]);
    sel->mode = 0.f; sel->denom = (float)st[1]; sel->kth = 0.f; sel->tie_w = 0.f;
  } else {                                   // loss.topk(n_min) (loss.py:326)
    const float kth = __uint_as_float(ws[2048]);
    const double need = n_min - st[3];       // how many of the tied losses belong to the top-k
    loss[0] = (float)((st[2] + need * (double)kth) / n_min);
    sel->mode = 1.f; sel->denom = (float)n_min; sel->kth = kth;
    sel->tie_w = st[4] > 0.0 ? (float)(need / st[4]) : 0.f;
  }
}
__global__ void ce_final_kernel(const double* acc, float* loss) { loss[0] = (float)(acc[0] / acc[1]); }


// ---- fast paths: dense channels-last logits ([N,H,W,C] contiguous, the layout Model.forward hands out).  A workgroup owns a
// strip of 256 consecutive pixels = 256*C contiguous elements, moved with 16-byte accesses through LDS. -------------------
constexpr int STRIP = 256;
// softmax pieces of one pixel row with the hardware exp2/log2 (v_exp_f32 / v_log_f32, ~1 ulp): the libm expf costs ~3x the
// issue slots and these kernels sit at the VALU/HBM balance point (19 exps per 38 bytes)
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

// GRAD: also leaves softmax - onehot (0 on ignored pixels) in `g` -- d(loss)/d(logits) up to the scalar gout/n_valid that
// only exists once the whole image set is reduced; the consumer applies it (myolo_seg_ce_scale + myolo_seg_upsample_bwd).
// The next strip (logits as 16-byte vectors + its target) is fetched into registers while the current one is reduced.
// CC: class count known at compile time (19 = Cityscapes, the reference's only use) -> branch-free unrolled rows; 0 = runtime C
template <typename T, bool GRAD, int CC>
__global__ __launch_bounds__(STRIP) void seg_ce_fwd_cl_kernel(const T* x, T* g, const int64_t* tgt, int Crt, int64_t total, int ignore,
                                                              double* acc, float* pix) {
  const int C = CC ? CC : Crt;
  constexpr int CMAX = CC ? CC : MAXC;
  __shared__ __attribute__((aligned(16))) T buf[STRIP * MAXC];
  __shared__ double sh[4];
  constexpr int V = 16 / (int)sizeof(T);
  constexpr int PRE = STRIP * MAXC / V / STRIP;            // 16-byte vectors per thread per full strip, upper bound
  const int nvec = STRIP * C / V;                           // dense_cl(): a full strip is a whole number of vectors
  double lsum = 0.0, lcnt = 0.0;
  const int64_t nstrips = (total + STRIP - 1) / STRIP;
  uint4 pre[PRE];
  int64_t tpre = ignore;
  auto fetch = [&](int64_t sidx) {
    const uint4* src = reinterpret_cast<const uint4*>(x + sidx * STRIP * C);
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      const int v = threadIdx.x + k * STRIP;
      if (k * STRIP < nvec) pre[k] = ldg16(src + (v < nvec ? v : nvec - 1));     // clamped: the loads stay unconditional
    }
    tpre = tgt[sidx * STRIP + threadIdx.x];
  };
  int64_t sidx = blockIdx.x;
  bool have = sidx < nstrips && (sidx + 1) * STRIP <= total;
  if (have) fetch(sidx);
  for (; sidx < nstrips; sidx += gridDim.x) {
    const int64_t p0 = sidx * STRIP;
    const int npix = total - p0 < STRIP ? (int)(total - p0) : STRIP;
    int64_t t = ignore;
    __syncthreads();
    if (have) {
#pragma unroll
      for (int k = 0; k < PRE; ++k) {
        const int v = threadIdx.x + k * STRIP;
        if (k * STRIP < nvec && v < nvec) reinterpret_cast<uint4*>(buf)[v] = pre[k];
      }
      t = tpre;
    } else {                                                // ragged last strip
      strip_load(x + p0 * C, buf, npix * C);
      if ((int)threadIdx.x < npix) t = tgt[p0 + threadIdx.x];
    }
    __syncthreads();
    const int64_t nxt = sidx + gridDim.x;
    have = nxt < nstrips && (nxt + 1) * STRIP <= total;
    if (have) fetch(nxt);
    if ((int)threadIdx.x < npix) {
      T* row = buf + threadIdx.x * C;
      float l = 0.f;
      if (t != ignore) {
        float v[CMAX];
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (CC || c < C) { v[c] = (float)row[c]; m = fmaxf(m, v[c]); }
        const float mb = -m * LOG2E;
        float sm = 0.f, xt = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (CC || c < C) {
            if (c == (int)t) xt = v[c];
            v[c] = __builtin_amdgcn_exp2f(fmaf(v[c], LOG2E, mb));
            sm += v[c];
          }
        l = (m + __builtin_amdgcn_logf(sm) * LN2) - xt;
        lsum += (double)l;
        lcnt += 1.0;
        if (GRAD) {
          const float inv = __builtin_amdgcn_rcpf(sm);
#pragma unroll
          for (int c = 0; c < CMAX; ++c)
            if (CC || c < C) row[c] = (T)(v[c] * inv - (c == (int)t ? 1.f : 0.f));
        }
      } else if (GRAD) {
#pragma unroll
        for (int c = 0; c < CMAX; ++c) if (CC || c < C) row[c] = (T)0.f;
      }
      if (pix) pix[p0 + threadIdx.x] = l;
    }
    if (GRAD) {
      __syncthreads();
      strip_store(g + p0 * C, buf, npix * C);
    }
  }
  const double bs = block_sum256(lsum, sh);
  const double bc = block_sum256(lcnt, sh);
  if (threadIdx.x == 0) { atomicAdd(acc + 0, bs); atomicAdd(acc + 1, bc); }
}
__global__ void ce_scale_kernel(const double* acc, const float* gout, float* scale) { scale[0] = (float)((double)gout[0] / acc[1]); }

template <typename T>
__global__ __launch_bounds__(STRIP) void seg_ce_bwd_cl_kernel(const T* x, T* g, const int64_t* tgt, int C, int64_t total,
                                                              int ignore, const double* acc, const float* gout,
                                                              const float* pix, const OhemSel* sel, float thresh) {
  __shared__ __attribute__((aligned(16))) T buf[STRIP * MAXC];
  const float go = gout[0];
  float wbase;
  int mode = -1;
  float kth = 0.f, tie_w = 0.f;
  if (sel) { mode = (int)sel->mode; kth = sel->kth; tie_w = sel->tie_w; wbase = go / sel->denom; }
  else wbase = (float)((double)go / acc[1]);
  const int64_t nstrips = (total + STRIP - 1) / STRIP;
  for (int64_t sidx = blockIdx.x; sidx < nstrips; sidx += gridDim.x) {
    const int64_t p0 = sidx * STRIP;
    const int npix = total - p0 < STRIP ? (int)(total - p0) : STRIP;
    __syncthreads();
    strip_load(x + p0 * C, buf, npix * C);
    __syncthreads();
    if ((int)threadIdx.x < npix) {
      const int64_t i = p0 + threadIdx.x;
      const int64_t t = tgt[i];
      float w = (t != ignore) ? wbase : 0.f;
      if (mode == 0) { if (!(pix[i] > thresh)) w = 0.f; }
      else if (mode == 1) { const float l = pix[i]; w = l > kth ? wbase : (l == kth ? wbase * tie_w : 0.f); if (t == ignore) w = 0.f; }
      T* row = buf + threadIdx.x * C;
      if (w == 0.f) {
#pragma unroll
        for (int c = 0; c < MAXC; ++c) if (c < C) row[c] = (T)0.f;
      } else {
        float v[MAXC];
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
          if (c < C) { v[c] = (float)row[c]; m = fmaxf(m, v[c]); }
        float sm = 0.f;
        const float mb = -m * LOG2E;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) if (c < C) { v[c] = __builtin_amdgcn_exp2f(fmaf(v[c], LOG2E, mb)); sm += v[c]; }
        const float inv = __builtin_amdgcn_rcpf(sm);
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
          if (c < C) row[c] = (T)((v[c] * inv - (c == (int)t ? 1.f : 0.f)) * w);
      }
    }
    __syncthreads();
    strip_store(g + p0 * C, buf, npix * C);
  }
}

// grid of a persistent strip loop: exactly the number of workgroups the device holds at once (no partial last wave of blocks)
inline int persistent_grid(const void* kern, int threads, int64_t work) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> cache;       // (device, kernel) -> resident workgroups
  int dev = 0;
  (void)hipGetDevice(&dev);
  int64_t g;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find({dev, kern});
    if (it == cache.end()) {
      int per_cu = 0, cus = 256;
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, 0) != hipSuccess || per_cu < 1) per_cu = 4;
      it = cache.emplace(std::make_pair(dev, kern), cus * per_cu).first;
    }
    g = it->second;
  }
  if (g > work) g = work;
  return g < 1 ? 1 : (int)g;
}

inline bool dense_cl(const void* p, int dt, int C, int H, int W, int64_t sn, int64_t sc, int64_t sh, int64_t sw) {
  const int es = dt == MYOLO_F16 ? 2 : 4;
  return (dt == MYOLO_F16 || dt == MYOLO_F32) && sc == 1 && sw == C && sh == (int64_t)W * C && sn == (int64_t)H * W * C &&
         (STRIP * C * es) % 16 == 0 && ((uintptr_t)p & 15) == 0 && C <= MAXC;
}

inline bool strided_ok(const void* p, int dt) { return p && (dt == MYOLO_F16 || dt == MYOLO_F32); }

}  // namespace

namespace {

// ---- K15: x`scale` bilinear upsample (align_corners=True) + mean cross entropy + its gradient + the transposed upsample in ONE pass over
// the LOW-resolution class logits (yolo.py:163 nn.Upsample + loss.py:236-237).  Unfused, the full-resolution logits are read back
// (38 B/pixel), softmax - onehot is written (38 B/pixel) and read again by the transposed upsample: ~1 GB per step at 16x512x1024.
// Here a workgroup owns 256 consecutive full-resolution columns x TY rows of one image; every thread walks its column downwards:
//   logits(y, x) = (1-ly)*top + ly*bot, top/bot = the two low rows around y interpolated in x ONCE per low row (registers),
//   rounded to the storage type like the materialised tensor the reference's loss reads; softmax / loss / gradient in registers;
//   the gradient is folded into the two low rows ((1-ly), ly) in registers; when the walk leaves a low row its per-column sums go
//   through LDS, each (low x, class) pair folds its <= 2/sx columns and adds the result to the fp32 low-res gradient (one global
//   atomic per pair, <= 4 workgroups touch an address).  HBM traffic: the int64 targets (8 B/pixel) + the low-res maps.
__device__ __forceinline__ void up_out_range(int i, int in, int out, float s, int& lo, int& hi) {
  if (out == 1 || in == 1 || s <= 0.f) { lo = 0; hi = out - 1; return; }
  lo = (int)floorf(((float)i - 1.f) / s) - 1; hi = (int)ceilf(((float)i + 1.f) / s) + 1;
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}

// MODE 0: mean CE -- loss sums + the unnormalised gradient (softmax - onehot) folded into the low-resolution map
// MODE 1: OhemCELoss first pass (loss.py:321-322): the per-pixel losses (`reduction='none'`: 0 on ignored pixels) to `pix`, the valid count
// MODE 2: OhemCELoss gradient pass: the same walk, every pixel's (softmax - onehot) weighted by the selection myolo_ohem_select left in
//         `sel` (loss > thresh, or the n_min largest with the tied ones sharing what is left, loss.py:323-327), folded into the low map
template <typename T, int CC, int MODE = 0>
__global__ __launch_bounds__(256) void seg_upce_kernel(myolo_tensor low, int H, int W, float sy, float sx, int TY, const int64_t* tgt,
                                                       int ignore, double* acc, float* g32, float* pix, const OhemSel* sel, float thresh) {
  constexpr int C = CC;
  int omode = 0; float okth = 0.f, otie = 0.f;
  if (MODE == 2) { omode = (int)sel->mode; okth = sel->kth; otie = sel->tie_w; }
  __shared__ float fb[256 * C];
  __shared__ double shd[4];
  const int strips = (W + 255) / 256, nyb = (H + TY - 1) / TY;
  int b = blockIdx.x;
  const int xs = b % strips; b /= strips;
  const int yb = b % nyb; const int n = b / nyb;
  const int xbase = xs * 256;
  const int x = xbase + threadIdx.x;
  const bool live = x < W;
  const int xc = live ? x : W - 1;
  const float fx = sx * (float)xc;
  const int x0 = (int)fx, x1 = x0 + 1 < low.w ? x0 + 1 : low.w - 1;
  const float lx = fx - (float)x0;
  const int xend = xbase + 255 < W - 1 ? xbase + 255 : W - 1;
  const int lxa = (int)(sx * (float)xbase);
  int lxz = (int)(sx * (float)xend) + 1;
  if (lxz > low.w - 1) lxz = low.w - 1;
  const int nlx = lxz - lxa + 1;
  const int ya = yb * TY, yz = (ya + TY < H ? ya + TY : H) - 1;
  const T* lp = reinterpret_cast<const T*>(low.ptr) + (int64_t)n * low.sn;
  float top[C], bot[C], at[C], ab[C];
  auto row_interp = [&](int r, float* dst) {
    const T* p0 = lp + (int64_t)r * low.sh + (int64_t)x0 * low.sw;
    const T* p1 = lp + (int64_t)r * low.sh + (int64_t)x1 * low.sw;
#pragma unroll
    for (int c = 0; c < C; ++c) dst[c] = (1.f - lx) * (float)p0[c] + lx * (float)p1[c];
  };
  // per-column sums `a` of low row r -> fp32 low-res gradient
  auto flush = [&](int r, const float* a) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C; ++c) fb[threadIdx.x * C + c] = live ? a[c] : 0.f;
    __syncthreads();
    for (int p = threadIdx.x; p < nlx * C; p += 256) {
      const int li = p / C, c = p - li * C;
      const int ix = lxa + li;
      int pxlo, pxhi;
      up_out_range(ix, low.w, W, sx, pxlo, pxhi);
      if (pxlo < xbase) pxlo = xbase;
      if (pxhi > xend) pxhi = xend;
      float sacc = 0.f;
      for (int ox = pxlo; ox <= pxhi; ++ox) {
        const float gx = sx * (float)ox; const int q0 = (int)gx; const int q1 = q0 + 1 < low.w ? q0 + 1 : low.w - 1;
        const float l = gx - (float)q0;
        float wx = 0.f;
        if (q0 == ix) wx += 1.f - l;
        if (q1 == ix) wx += l;
        sacc += wx * fb[(ox - xbase) * C + c];
      }
      if (sacc != 0.f) atomicAdd(g32 + (((int64_t)n * low.h + r) * low.w + ix) * C + c, sacc);
    }
  };
  int j = (int)(sy * (float)ya);
  row_interp(j, top);
  row_interp(j + 1 < low.h ? j + 1 : low.h - 1, bot);
#pragma unroll
  for (int c = 0; c < C; ++c) { at[c] = 0.f; ab[c] = 0.f; }
  double lsum = 0.0, lcnt = 0.0;
  const int64_t* tp = tgt + ((int64_t)n * H + ya) * W + xc;
  int64_t tnext = live ? tp[0] : (int64_t)ignore;
  for (int y = ya; y <= yz; ++y) {
    const int64_t t = tnext;
    if (y < yz && live) tnext = tp[(int64_t)(y + 1 - ya) * W];          // next row's target in flight
    const float fy = sy * (float)y;
    const int y0 = (int)fy;
    const float ly = fy - (float)y0;
    while (j != y0) {                                   // uniform over the workgroup: the walk left low row j
      if (MODE != 1) flush(j, at);
      ++j;
#pragma unroll
      for (int c = 0; c < C; ++c) { at[c] = ab[c]; ab[c] = 0.f; top[c] = bot[c]; }
      row_interp(j + 1 < low.h ? j + 1 : low.h - 1, bot);
    }
    if (t != ignore) {
      float v[C];
      float m = -INFINITY;
#pragma unroll
      for (int c = 0; c < C; ++c) { v[c] = (float)(T)((1.f - ly) * top[c] + ly * bot[c]); m = fmaxf(m, v[c]); }
      const float mb = -m * LOG2E;
      float sm = 0.f, xt = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        if (c == (int)t) xt = v[c];
        v[c] = __builtin_amdgcn_exp2f(fmaf(v[c], LOG2E, mb));
        sm += v[c];
      }
      const float l = (m + __builtin_amdgcn_logf(sm) * LN2) - xt;
      lsum += (double)l;
      lcnt += 1.0;
      if (MODE == 1) {
        if (live) pix[((int64_t)n * H + y) * W + x] = l;
      } else {
        float wgt = 1.f;
        if (MODE == 2) {                                // the loss value the selection was made on (bit-identical: written by MODE 1)
          const float lp = pix[((int64_t)n * H + y) * W + xc];
          wgt = omode == 0 ? (lp > thresh ? 1.f : 0.f) : (lp > okth ? 1.f : (lp == okth ? otie : 0.f));
        }
        const float inv = __builtin_amdgcn_rcpf(sm) * wgt;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float g = v[c] * inv - (c == (int)t ? wgt : 0.f);
          at[c] += (1.f - ly) * g;
          ab[c] += ly * g;
        }
      }
    } else if (MODE == 1 && live) {
      pix[((int64_t)n * H + y) * W + x] = 0.f;
    }
  }
  if (MODE != 1) {
    flush(j, at);
    flush(j + 1 < low.h ? j + 1 : low.h - 1, ab);
  }
  if (MODE != 2) {
    const double bs = block_sum256(lsum, shd);
    const double bc = block_sum256(lcnt, shd);
    if (threadIdx.x == 0) { atomicAdd(acc + 0, bs); atomicAdd(acc + 1, bc); }
  }
}

// low-res gradient of the head's classifier: glow (+)= scale * g32   (scale = gout / n_valid, myolo_seg_ce_scale)
template <typename T>
__global__ __launch_bounds__(256) void seg_lowgrad_apply_kernel(const float* __restrict__ g32, myolo_tensor glow, int C, int acc,
                                                                const float* scale) {
  const float gs = scale ? scale[0] : 1.f;
  const int64_t total = (int64_t)glow.n * glow.h * glow.w * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    const int c = (int)(r % C); r /= C;
    const int ix = (int)(r % glow.w); r /= glow.w;
    const int iy = (int)(r % glow.h); const int n = (int)(r / glow.h);
    T* o = reinterpret_cast<T*>(glow.ptr) + (int64_t)n * glow.sn + (int64_t)iy * glow.sh + (int64_t)ix * glow.sw + c;
    float v = g32[i] * gs;
    if (acc) v += (float)*o;
    *o = (T)v;
  }
}

}  // namespace

extern "C" int myolo_seg_upce_fwd_grad(const myolo_tensor* low, int H, int W, const int64_t* target, int ignore_index, double* acc,
                                       float* loss, float* glow32, void* stream) {
  if (!low || !low->ptr || !target || !acc || !glow32 || H < 1 || W < 1) return MYOLO_EINVAL;
  if (low->dtype != MYOLO_F16 && low->dtype != MYOLO_F32) return MYOLO_EINVAL;
  if (low->c != 19) return MYOLO_EINVAL;                       // Cityscapes' 19 classes (the reference's only use): other counts take the unfused path
  hipStream_t st = (hipStream_t)stream;
  int e = myolo_fill_words2(acc, 2 * sizeof(double), 0u, glow32, (size_t)low->n * low->h * low->w * low->c * sizeof(float), 0u, st);
  if (e) return e;
  const float sy = H > 1 ? (float)(low->h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(low->w - 1) / (float)(W - 1) : 0.f;
  int TY = 32;
  const int strips = (W + 255) / 256;
  while (TY > 8 && (int64_t)low->n * ((H + TY - 1) / TY) * strips < 1024) TY >>= 1;     // >= 4 workgroups per CU when the map allows
  const int64_t grid = (int64_t)low->n * ((H + TY - 1) / TY) * strips;
  if (grid > 0x7fffffff) return MYOLO_EINVAL;
  if (low->dtype == MYOLO_F16)
    hipLaunchKernelGGL((seg_upce_kernel<half_t, 19, 0>), dim3((unsigned)grid), dim3(256), 0, st, *low, H, W, sy, sx, TY, target, ignore_index,
                       acc, glow32, (float*)nullptr, (const OhemSel*)nullptr, 0.f);
  else
    hipLaunchKernelGGL((seg_upce_kernel<float, 19, 0>), dim3((unsigned)grid), dim3(256), 0, st, *low, H, W, sy, sx, TY, target, ignore_index,
                       acc, glow32, (float*)nullptr, (const OhemSel*)nullptr, 0.f);
  MYOLO_CHECK_LAUNCH();
  if (loss) {
    hipLaunchKernelGGL(ce_final_kernel, dim3(1), dim3(1), 0, st, acc, loss);
    MYOLO_CHECK_LAUNCH();
  }
  return 0;
}

// OhemCELoss over the LOW-resolution logits (SURVEY K15 for loss.py:303-328): pass 1 = per-pixel losses (the only full-resolution
// tensor that exists: fp32 [n][H][W]) + valid count; myolo_ohem_select picks the hard pixels; pass 2 = their gradient, folded into the
// low-resolution map.  The x8-upsampled logits and their gradient are never formed.
static int seg_upce_ohem_launch(const myolo_tensor* low, int H, int W, const int64_t* target, int ignore_index, double* acc, float* pix,
                                const float* sel, float thresh, float* glow32, int mode, void* stream) {
  if (!low || !low->ptr || !target || !pix || H < 1 || W < 1) return MYOLO_EINVAL;
  if ((low->dtype != MYOLO_F16 && low->dtype != MYOLO_F32) || low->c != 19) return MYOLO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (mode == 1) {
    if (!acc) return MYOLO_EINVAL;
    const int ef = myolo_fill_words2(acc, 2 * sizeof(double), 0u, nullptr, 0, 0u, st);
    if (ef) return ef;
  } else {
    if (!sel || !glow32) return MYOLO_EINVAL;
    const int ef = myolo_fill_words2(glow32, (size_t)low->n * low->h * low->w * low->c * sizeof(float), 0u, nullptr, 0, 0u, st);
    if (ef) return ef;
  }
  const float sy = H > 1 ? (float)(low->h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(low->w - 1) / (float)(W - 1) : 0.f;
  int TY = 32;
  const int strips = (W + 255) / 256;
  while (TY > 8 && (int64_t)low->n * ((H + TY - 1) / TY) * strips < 1024) TY >>= 1;
  const int64_t grid = (int64_t)low->n * ((H + TY - 1) / TY) * strips;
  if (grid > 0x7fffffff) return MYOLO_EINVAL;
  const OhemSel* os = reinterpret_cast<const OhemSel*>(sel);
  if (low->dtype == MYOLO_F16) {
    if (mode == 1) hipLaunchKernelGGL((seg_upce_kernel<half_t, 19, 1>), dim3((unsigned)grid), dim3(256), 0, st, *low, H, W, sy, sx, TY, target, ignore_index, acc, glow32, pix, os, thresh);
    else hipLaunchKernelGGL((seg_upce_kernel<half_t, 19, 2>), dim3((unsigned)grid), dim3(256), 0, st, *low, H, W, sy, sx, TY, target, ignore_index, acc, glow32, pix, os, thresh);
  } else {
    if (mode == 1) hipLaunchKernelGGL((seg_upce_kernel<float, 19, 1>), dim3((unsigned)grid), dim3(256), 0, st, *low, H, W, sy, sx, TY, target, ignore_index, acc, glow32, pix, os, thresh);
    else hipLaunchKernelGGL((seg_upce_kernel<float, 19, 2>), dim3((unsigned)grid), dim3(256), 0, st, *low, H, W, sy, sx, TY, target, ignore_index, acc, glow32, pix, os, thresh);
  }
  MYOLO_CHECK_LAUNCH();
  return 0;
}
extern "C" int myolo_seg_upce_ohem_pix(const myolo_tensor* low, int H, int W, const int64_t* target, int ignore_index, double* acc,
                                       float* pix, void* stream) {
  return seg_upce_ohem_launch(low, H, W, target, ignore_index, acc, pix, nullptr, 0.f, nullptr, 1, stream);
}
extern "C" int myolo_seg_upce_ohem_grad(const myolo_tensor* low, int H, int W, const int64_t* target, int ignore_index, const float* pix,
                                        const float* sel, float thresh, float* glow32, void* stream) {
  return seg_upce_ohem_launch(low, H, W, target, ignore_index, nullptr, const_cast<float*>(pix), sel, thresh, glow32, 2, stream);
}

extern "C" int myolo_seg_lowgrad_apply(const float* glow32, const myolo_tensor* glow, int accumulate, const float* scale, void* stream) {
  if (!glow32 || !glow || !glow->ptr || (glow->dtype != MYOLO_F16 && glow->dtype != MYOLO_F32)) return MYOLO_EINVAL;
  const int64_t total = (int64_t)glow->n * glow->h * glow->w * glow->c;
  const int grid = grid_for(total, 256, 4096);
  if (glow->dtype == MYOLO_F16)
    hipLaunchKernelGGL(seg_lowgrad_apply_kernel<half_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, glow32, *glow, glow->c, accumulate, scale);
  else
    hipLaunchKernelGGL(seg_lowgrad_apply_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, glow32, *glow, glow->c, accumulate, scale);
  MYOLO_CHECK_LAUNCH();
  return 0;
}

static int seg_ce_fwd_impl(const void* logits, void* grad, int dtype, int n, int c, int h, int w, int64_t sn, int64_t sc, int64_t sh,
                           int64_t sw, const int64_t* target, int ignore_index, double* acc, float* pix, float* loss,
                           void* stream) {
  if (!strided_ok(logits, dtype) || !target || !acc || c < 1 || c > MAXC || n < 1 || h < 1 || w < 1) return MYOLO_EINVAL;
  const bool dense = dense_cl(logits, dtype, c, h, w, sn, sc, sh, sw);
  if (grad && (!dense || ((uintptr_t)grad & 15))) return MYOLO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int e = myolo_fill_words2(acc, 2 * sizeof(double), 0u, nullptr, 0, 0u, st);
  if (e) return e;
  const int64_t total = (int64_t)n * h * w;
  Strided4 x{const_cast<void*>(logits), sn, sc, sh, sw, dtype};
  if (dense) {                                                     // all pixels of all images are one contiguous [P][C] array
    auto launch = [&](auto kern, auto* xp, auto* gp) {
      const int grid = persistent_grid(reinterpret_cast<const void*>(kern), STRIP, (total + STRIP - 1) / STRIP);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(STRIP), 0, st, xp, gp, target, c, total, ignore_index, acc, pix);
    };
    if (dtype == MYOLO_F16) {
      const half_t* xp = (const half_t*)logits; half_t* gp = (half_t*)grad;
      if (c == 19) { if (grad) launch(seg_ce_fwd_cl_kernel<half_t, true, 19>, xp, gp); else launch(seg_ce_fwd_cl_kernel<half_t, false, 19>, xp, gp); }
      else { if (grad) launch(seg_ce_fwd_cl_kernel<half_t, true, 0>, xp, gp); else launch(seg_ce_fwd_cl_kernel<half_t, false, 0>, xp, gp); }
    } else {
      const float* xp = (const float*)logits; float* gp = (float*)grad;
      if (c == 19) { if (grad) launch(seg_ce_fwd_cl_kernel<float, true, 19>, xp, gp); else launch(seg_ce_fwd_cl_kernel<float, false, 19>, xp, gp); }
      else { if (grad) launch(seg_ce_fwd_cl_kernel<float, true, 0>, xp, gp); else launch(seg_ce_fwd_cl_kernel<float, false, 0>, xp, gp); }
    }
  } else {
    hipLaunchKernelGGL(seg_ce_fwd_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, st, x, target, c, h, w, total,
                       ignore_index, acc, pix);
  }
  MYOLO_CHECK_LAUNCH();
  if (loss) {
    hipLaunchKernelGGL(ce_final_kernel, dim3(1), dim3(1), 0, st, acc, loss);
    MYOLO_CHECK_LAUNCH();
  }
  return 0;
}
extern "C" int myolo_seg_ce_fwd(const void* logits, int dtype, int n, int c, int h, int w, int64_t sn, int64_t sc, int64_t sh,
                                int64_t sw, const int64_t* target, int ignore_index, double* acc, float* pix, float* loss,
                                void* stream) {
  return seg_ce_fwd_impl(logits, nullptr, dtype, n, c, h, w, sn, sc, sh, sw, target, ignore_index, acc, pix, loss, stream);
}
extern "C" int myolo_seg_ce_fwd_grad(const void* logits, void* grad, int dtype, int n, int c, int h, int w,
                                     const int64_t* target, int ignore_index, double* acc, float* loss, void* stream) {
  if (!grad) return MYOLO_EINVAL;
  return seg_ce_fwd_impl(logits, grad, dtype, n, c, h, w, (int64_t)h * w * c, 1, (int64_t)w * c, c, target, ignore_index, acc,
                         nullptr, loss, stream);
}
extern "C" int myolo_seg_ce_scale(const double* acc, const float* gout, float* scale, void* stream) {
  if (!acc || !gout || !scale) return MYOLO_EINVAL;
  hipLaunchKernelGGL(ce_scale_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, acc, gout, scale);
  MYOLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int myolo_ohem_select(const float* pix, int64_t total, float thresh, const double* acc, double* st,
                                 uint32_t* ws, float* loss, float* sel, void* stream) {
  if (!pix || !acc || !st || !ws || !loss || !sel || total < 1) return MYOLO_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int e = myolo_fill_words2(st, 5 * sizeof(double), 0u, ws, 2052 * sizeof(uint32_t), 0u, s);
  if (e) return e;
  const int grid = grid_for(total, 256, 2048);
  hipLaunchKernelGGL(ohem_thresh_kernel, dim3(grid), dim3(256), 0, s, pix, total, thresh, st);
  // 32-bit keys in passes of 11 + 11 + 10 bits
  const int shifts[3] = {21, 10, 0}, nbs[3] = {11, 11, 10};
  for (int p = 0; p < 3; ++p) {
    const uint32_t himask = p == 0 ? 0u : (0xffffffffu << (shifts[p] + nbs[p]));
    hipLaunchKernelGGL(ohem_hist_kernel, dim3(grid), dim3(256), 0, s, pix, total, ws, shifts[p], nbs[p], himask);
    hipLaunchKernelGGL(ohem_pick_kernel, dim3(1), dim3(256), 0, s, ws, shifts[p], nbs[p], acc, p == 0 ? 1 : 0);
  }
  hipLaunchKernelGGL(ohem_topk_sum_kernel, dim3(grid), dim3(256), 0, s, pix, total, ws, st);
  hipLaunchKernelGGL(ohem_final_kernel, dim3(1), dim3(1), 0, s, acc, st, ws, loss, reinterpret_cast<OhemSel*>(sel));
  MYOLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int myolo_seg_ce_bwd(const void* logits, void* grad, int dtype, int n, int c, int h, int w, int64_t sn, int64_t sc,
                                int64_t sh, int64_t sw, int64_t gsn, int64_t gsc, int64_t gsh, int64_t gsw,
                                const int64_t* target, int ignore_index, const double* acc, const float* gout,
                                const float* pix, const float* sel, float thresh, void* stream) {
  if (!strided_ok(logits, dtype) || !grad || !target || !acc || !gout || c < 1 || c > MAXC) return MYOLO_EINVAL;
  if ((sel != nullptr) != (pix != nullptr)) return MYOLO_EINVAL;
  const int64_t total = (int64_t)n * h * w;
  Strided4 x{const_cast<void*>(logits), sn, sc, sh, sw, dtype}, g{grad, gsn, gsc, gsh, gsw, dtype};
  if (dense_cl(logits, dtype, c, h, w, sn, sc, sh, sw) && dense_cl(grad, dtype, c, h, w, gsn, gsc, gsh, gsw)) {
    const int grid = grid_for(total, STRIP, 4096);
    if (dtype == MYOLO_F16)
      hipLaunchKernelGGL(seg_ce_bwd_cl_kernel<half_t>, dim3(grid), dim3(STRIP), 0, (hipStream_t)stream, (const half_t*)logits, (half_t*)grad,
                         target, c, total, ignore_index, acc, gout, pix, reinterpret_cast<const OhemSel*>(sel), thresh);
    else
      hipLaunchKernelGGL(seg_ce_bwd_cl_kernel<float>, dim3(grid), dim3(STRIP), 0, (hipStream_t)stream, (const float*)logits, (float*)grad,
                         target, c, total, ignore_index, acc, gout, pix, reinterpret_cast<const OhemSel*>(sel), thresh);
  } else {
    hipLaunchKernelGGL(seg_ce_bwd_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, g, target, c,
                       h, w, total, ignore_index, acc, gout, pix, reinterpret_cast<const OhemSel*>(sel), thresh);
  }
  MYOLO_CHECK_LAUNCH();
  return 0;
}

// ================================================================================================ detection
namespace {

struct DetK {
  int nl, na, no, bs, nt, dtype;
  const void* p[5]; void* gp[5];
  int ny[5], nx[5];
  int64_t cell0[5];          // first cell of level l in the concatenated cell index space
  const float* anchors; const float* targets;
  float balance[5];
  float box, obj, cls, cls_pw, obj_pw, anchor_t, gr, cp, cn;
  int* winner; float* ciou; double* acc; float* out; float* gp32; const float* gout;
};

// forward-mode dual number over the 4 predicted box parameters (x, y, w, h)
struct D4 {
  float v, d[4];
};
__device__ __forceinline__ D4 dc(float c) { return D4{c, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ D4 operator+(const D4& a, const D4& b) { return D4{a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2], a.d[3] + b.d[3]}}; }
__device__ __forceinline__ D4 operator-(const D4& a, const D4& b) { return D4{a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2], a.d[3] - b.d[3]}}; }
__device__ __forceinline__ D4 operator*(const D4& a, const D4& b) {
  D4 r; r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
__device__ __forceinline__ D4 operator/(const D4& a, const D4& b) {
  D4 r; r.v = a.v / b.v;
  const float inv = 1.f / b.v;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
  return r;
}
__device__ __forceinline__ D4 operator*(const D4& a, float s) { return D4{a.v * s, {a.d[0] * s, a.d[1] * s, a.d[2] * s, a.d[3] * s}}; }
__device__ __forceinline__ D4 operator+(const D4& a, float s) { D4 r = a; r.v += s; return r; }
__device__ __forceinline__ D4 dmin(const D4& a, const D4& b) { return a.v <= b.v ? a : b; }   // torch.min/max: tie -> first arg's value; grad split ignored (measure-zero)
__device__ __forceinline__ D4 dmax(const D4& a, const D4& b) { return a.v >= b.v ? a : b; }
__device__ __forceinline__ D4 dclamp0(const D4& a) { return a.v > 0.f ? a : (a.v == 0.f ? D4{0.f, {a.d[0], a.d[1], a.d[2], a.d[3]}} : dc(0.f)); }
__device__ __forceinline__ D4 datan(const D4& a) {
  const float g = 1.f / (1.f + a.v * a.v);
  return D4{atanf(a.v), {a.d[0] * g, a.d[1] * g, a.d[2] * g, a.d[3] * g}};
}

// bbox_iou(box1=pbox, box2=tbox, x1y1x2y2=False, CIoU=True) (general.py:343-380), gradients w.r.t. pbox; alpha is constant
// (computed under torch.no_grad, general.py:378-379)
__device__ __forceinline__ D4 ciou_dual(const D4 px, const D4 py, const D4 pw, const D4 ph, float tx, float ty, float tw, float th) {
  const float eps = 1e-7f;
  const D4 b1x1 = px - pw * 0.5f, b1x2 = px + pw * 0.5f, b1y1 = py - ph * 0.5f, b1y2 = py + ph * 0.5f;
  const float b2x1 = tx - tw / 2, b2x2 = tx + tw / 2, b2y1 = ty - th / 2, b2y2 = ty + th / 2;
  const D4 iw = dclamp0(dmin(b1x2, dc(b2x2)) - dmax(b1x1, dc(b2x1)));
  const D4 ih = dclamp0(dmin(b1y2, dc(b2y2)) - dmax(b1y1, dc(b2y1)));
  const D4 inter = iw * ih;
  const D4 w1 = b1x2 - b1x1, h1 = (b1y2 - b1y1) + eps;
  const float w2 = b2x2 - b2x1, h2 = b2y2 - b2y1 + eps;
  const D4 uni = ((w1 * h1 + w2 * h2) - inter) + eps;
  const D4 iou = inter / uni;
  const D4 cw = dmax(b1x2, dc(b2x2)) - dmin(b1x1, dc(b2x1));
  const D4 ch = dmax(b1y2, dc(b2y2)) - dmin(b1y1, dc(b2y1));
  const D4 c2 = (cw * cw + ch * ch) + eps;
  const D4 dxs = (dc(b2x1 + b2x2) - b1x1) - b1x2, dys = (dc(b2y1 + b2y2) - b1y1) - b1y2;
  const D4 rho2 = (dxs * dxs + dys * dys) * 0.25f;
  const D4 at = dc(atanf(w2 / h2)) - datan(w1 / h1);
  const D4 v = (at * at) * (4.f / (3.14159265358979323846f * 3.14159265358979323846f));
  const float alpha = v.v / (v.v - iou.v + (1.f + eps));
  return iou - (rho2 / c2 + v * alpha);
}

__device__ __forceinline__ float bce_logits(float x, float t, float pw) {
  // binary_cross_entropy_with_logits with pos_weight: (1-t)*x + (1+(pw-1)*t) * (log1p(exp(-|x|)) + max(-x,0))
  const float lw = 1.f + (pw - 1.f) * t;
  return (1.f - t) * x + lw * (log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.f));
}
__device__ __forceinline__ float bce_logits_grad(float x, float t, float pw) {
  const float lw = 1.f + (pw - 1.f) * t;
  const float s = 1.f / (1.f + expf(-x));
  return (1.f - t) - lw * (1.f - s);
}

struct Cand {
  bool valid; int b, a, cls, gj, gi; float tx, ty, tw, th, aw, ah;
};
// candidate (o, a, t) of level l in the reference's row order: offset-major, then anchor, then target (loss.py:170,189,197)
__device__ __forceinline__ Cand make_cand(const DetK& k, int l, int idx) {
  Cand c;
  const int per_o = k.na * k.nt;
  const int o = idx / per_o; const int r = idx - o * per_o; const int a = r / k.nt; const int t = r - a * k.nt;
  const float* tg = k.targets + (int64_t)t * 6;
  const float nx = (float)k.nx[l], ny = (float)k.ny[l];
  const float gx = tg[2] * nx, gy = tg[3] * ny, gw = tg[4] * nx, gh = tg[5] * ny;
  const float aw = k.anchors[(l * k.na + a) * 2], ah = k.anchors[(l * k.na + a) * 2 + 1];
  const float rw = gw / aw, rh = gh / ah;
  const float mr = fmaxf(fmaxf(rw, 1.f / rw), fmaxf(rh, 1.f / rh));
  bool ok = mr < k.anchor_t;                                          // loss.py:186-187
  float ox = 0.f, oy = 0.f;
  if (o == 1) { ok = ok && (gx - floorf(gx) < 0.5f) && gx > 1.f; ox = 0.5f; }
  else if (o == 2) { ok = ok && (gy - floorf(gy) < 0.5f) && gy > 1.f; oy = 0.5f; }
  else if (o == 3) { const float q = nx - gx; ok = ok && (q - floorf(q) < 0.5f) && q > 1.f; ox = -0.5f; }
  else if (o == 4) { const float q = ny - gy; ok = ok && (q - floorf(q) < 0.5f) && q > 1.f; oy = -0.5f; }
  c.valid = ok;
  int gi = (int)(gx - ox), gj = (int)(gy - oy);                        // .long() truncates toward zero (loss.py:206)
  gi = gi < 0 ? 0 : (gi > k.nx[l] - 1 ? k.nx[l] - 1 : gi);             // clamp_ in place (loss.py:212) -> tbox sees it too
  gj = gj < 0 ? 0 : (gj > k.ny[l] - 1 ? k.ny[l] - 1 : gj);
  c.gi = gi; c.gj = gj;
  c.b = (int)tg[0]; c.cls = (int)tg[1]; c.a = a;
  c.tx = gx - (float)gi; c.ty = gy - (float)gj; c.tw = gw; c.th = gh;
  c.aw = aw; c.ah = ah;
  if (c.b < 0 || c.b >= k.bs) c.valid = false;
  return c;
}
__device__ __forceinline__ int64_t cell_of(const DetK& k, int l, const Cand& c) {
  return (((int64_t)c.b * k.na + c.a) * k.ny[l] + c.gj) * k.nx[l] + c.gi;
}

// pass A: per candidate CIoU / box + class loss sums / objectness-target winner
__global__ __launch_bounds__(256) void det_cand_fwd_kernel(const DetK k) {
  __shared__ double sh[4];
  const int l = blockIdx.y;
  const int ncand = 5 * k.na * k.nt;
  double sbox = 0.0, scnt = 0.0, scls = 0.0;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < ncand; idx += gridDim.x * blockDim.x) {
    const Cand c = make_cand(k, l, idx);
    float ci = 0.f;
    if (c.valid) {
      const int64_t cell = cell_of(k, l, c);
      const int64_t row = cell * k.no;
      const float s0 = sigmoid_f(ld_any(k.p[l], row + 0, k.dtype)), s1 = sigmoid_f(ld_any(k.p[l], row + 1, k.dtype));
      const float s2 = sigmoid_f(ld_any(k.p[l], row + 2, k.dtype)), s3 = sigmoid_f(ld_any(k.p[l], row + 3, k.dtype));
      const float px = s0 * 2.f - 0.5f, py = s1 * 2.f - 0.5f;
      const float pw = (s2 * 2.f) * (s2 * 2.f) * c.aw, ph = (s3 * 2.f) * (s3 * 2.f) * c.ah;
      ci = ciou_dual(dc(px), dc(py), dc(pw), dc(ph), c.tx, c.ty, c.tw, c.th).v;
      sbox += (double)(1.f - ci);
      scnt += 1.0;
      const int nc = k.no - 5;
      if (nc > 1)
        for (int j = 0; j < nc; ++j)
          scls += (double)bce_logits(ld_any(k.p[l], row + 5 + j, k.dtype), j == c.cls ? k.cp : k.cn, k.cls_pw);
      atomicMax(k.winner + k.cell0[l] + cell, idx);
    }
    k.ciou[(int64_t)l * ncand + idx] = ci;
  }
  const double a = block_sum256(sbox, sh), b = block_sum256(scnt, sh), c2 = block_sum256(scls, sh);
  if (threadIdx.x == 0) { atomicAdd(k.acc + l * 4 + 0, a); atomicAdd(k.acc + l * 4 + 1, b); atomicAdd(k.acc + l * 4 + 2, c2); }
}

__device__ __forceinline__ int level_of(const DetK& k, int64_t cell, int64_t& local) {
  int l = 0;
  while (l + 1 < k.nl && cell >= k.cell0[l + 1]) ++l;
  local = cell - k.cell0[l];
  return l;
}
__device__ __forceinline__ float tobj_of(const DetK& k, int l, int64_t cell_global) {
  const int wi = k.winner[cell_global];
  if (wi < 0) return 0.f;
  const float ci = k.ciou[(int64_t)l * (5 * k.na * k.nt) + wi];
  return (1.f - k.gr) + k.gr * fmaxf(ci, 0.f);                        // loss.py:137
}

// pass B: objectness BCE over every cell of every level
__global__ __launch_bounds__(256) void det_obj_fwd_kernel(const DetK k, int64_t ncell) {
  __shared__ double sh[4];
  double s[5] = {0, 0, 0, 0, 0};
  // four cells per thread and step, their logit and winner loads issued together (round 6: one cell per step kept each 2-byte strided load next
  // to its wait: 23 us for 516 k cells)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncell; i += 4 * stride) {
    float x[4];
    int wi[4], lv[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t ii = i + u * stride;
      ok[u] = ii < ncell;
      const int64_t ic = ok[u] ? ii : ncell - 1;
      int64_t local;
      lv[u] = level_of(k, ic, local);
      x[u] = ld_any(k.p[lv[u]], local * k.no + 4, k.dtype);
      wi[u] = k.winner[ic];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float t = 0.f;
      if (wi[u] >= 0) t = (1.f - k.gr) + k.gr * fmaxf(k.ciou[(int64_t)lv[u] * (5 * k.na * k.nt) + wi[u]], 0.f);      // tobj_of (loss.py:137)
      if (ok[u]) s[lv[u]] += (double)bce_logits(x[u], t, k.obj_pw);
    }
  }
  for (int l = 0; l < k.nl; ++l) {
    const double v = block_sum256(s[l], sh);
    if (threadIdx.x == 0 && v != 0.0) atomicAdd(k.acc + l * 4 + 3, v);
  }
}

__global__ void det_final_kernel(const DetK k) {
  double lbox = 0, lobj = 0, lcls = 0;
  const int nc = k.no - 5;
  for (int l = 0; l < k.nl; ++l) {
    const double n = k.acc[l * 4 + 1];
    if (n > 0) {
      lbox += k.acc[l * 4 + 0] / n;
      if (nc > 1) lcls += k.acc[l * 4 + 2] / (n * nc);
    }
    const double cells = (double)k.bs * k.na * k.ny[l] * k.nx[l];
    lobj += k.acc[l * 4 + 3] / cells * (double)k.balance[l];
  }
  lbox *= k.box; lobj *= k.obj; lcls *= k.cls;
  const double loss = lbox + lobj + lcls;
  k.out[0] = (float)(loss * k.bs); k.out[1] = (float)lbox; k.out[2] = (float)lobj; k.out[3] = (float)lcls; k.out[4] = (float)loss;
}

// backward pass 1: every cell row = zeros except the objectness logit gradient
template <typename GT>
__global__ __launch_bounds__(256) void det_obj_bwd_kernel(const DetK k, int64_t ncell, GT* const g0, GT* const g1, GT* const g2,
                                                          GT* const g3, GT* const g4) {
  const float go = k.gout[0] * (float)k.bs;
  GT* const gs[5] = {g0, g1, g2, g3, g4};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncell; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t local;
    const int l = level_of(k, i, local);
    const float x = ld_any(k.p[l], local * k.no + 4, k.dtype);
    const float cells = (float)k.bs * k.na * k.ny[l] * k.nx[l];
    const float g = bce_logits_grad(x, tobj_of(k, l, i), k.obj_pw) * (go * k.obj * k.balance[l] / cells);
    // round 6: a wave's 64 cells are 64 * no CONSECUTIVE elements of the level's gradient buffer -- written as `no` coalesced 256-byte stores
    // (element e of the region: cell e / no, column e % no, the cell's value through a shuffle) instead of 64 rows of `no` scalar stores 4 * no
    // bytes apart (30 us for 31 MB).  A wave that straddles a level boundary or the end keeps the row form.
    const int lane = threadIdx.x & 63;
    const int64_t i0 = i - lane;
    int64_t l0, l63;
    const bool whole = i0 + 63 < ncell && level_of(k, i0, l0) == l && level_of(k, i0 + 63, l63) == l;
    if (__all(whole)) {
      GT* reg = gs[l] + (local - lane) * k.no;
      for (int r = 0; r < k.no; ++r) {
        const int e = r * 64 + lane;
        const int c = e / k.no, j = e - c * k.no;
        const float v = __shfl(g, c, 64);
        reg[e] = (GT)(j == 4 ? v : 0.f);
      }
    } else {
      GT* row = gs[l] + local * k.no;
      for (int j = 0; j < k.no; ++j) row[j] = (GT)(j == 4 ? g : 0.f);
    }
  }
}
// backward pass 2: box + class gradients of the matched cells (several candidates may hit one cell -> fp32 atomics)
__global__ __launch_bounds__(256) void det_cand_bwd_kernel(const DetK k, float* const g0, float* const g1, float* const g2,
                                                           float* const g3, float* const g4) {
  const int l = blockIdx.y;
  float* const gs[5] = {g0, g1, g2, g3, g4};
  const int ncand = 5 * k.na * k.nt;
  const double n = k.acc[l * 4 + 1];
  if (n <= 0) return;
  const float go = k.gout[0] * (float)k.bs;
  const int nc = k.no - 5;
  const float wbox = -go * k.box / (float)n;                 // d(1 - ciou).mean()
  const float wcls = nc > 1 ? go * k.cls / (float)(n * nc) : 0.f;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < ncand; idx += gridDim.x * blockDim.x) {
    const Cand c = make_cand(k, l, idx);
    if (!c.valid) continue;
    const int64_t row = cell_of(k, l, c) * k.no;
    float s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = sigmoid_f(ld_any(k.p[l], row + j, k.dtype));
    D4 px = dc(s[0] * 2.f - 0.5f), py = dc(s[1] * 2.f - 0.5f);
    D4 pw = dc((s[2] * 2.f) * (s[2] * 2.f) * c.aw), ph = dc((s[3] * 2.f) * (s[3] * 2.f) * c.ah);
    px.d[0] = 1.f; py.d[1] = 1.f; pw.d[2] = 1.f; ph.d[3] = 1.f;
    const D4 ci = ciou_dual(px, py, pw, ph, c.tx, c.ty, c.tw, c.th);
    // chain through the decode: dpx/dx = 2 s(1-s); dpw/dx = 8 s^2 (1-s) * anchor
    const float dd[4] = {2.f * s[0] * (1.f - s[0]), 2.f * s[1] * (1.f - s[1]),
                         8.f * s[2] * s[2] * (1.f - s[2]) * c.aw, 8.f * s[3] * s[3] * (1.f - s[3]) * c.ah};
    float* g = gs[l] + row;
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(g + j, wbox * ci.d[j] * dd[j]);
    if (nc > 1)
      for (int j = 0; j < nc; ++j)
        atomicAdd(g + 5 + j, wcls * bce_logits_grad(ld_any(k.p[l], row + 5 + j, k.dtype), j == c.cls ? k.cp : k.cn, k.cls_pw));
  }
}
__global__ __launch_bounds__(256) void cast_f32_to_f16_kernel(const float* src, half_t* dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = (half_t)src[i];
}

int fill_detk(const myolo_detloss_desc* d, DetK& k, int64_t& ncell) {
  if (!d || d->nl < 1 || d->nl > 5 || d->na < 1 || d->no < 5 || d->no - 5 > 256 || d->bs < 1 || d->nt < 0) return MYOLO_EINVAL;
  if (d->dtype != MYOLO_F16 && d->dtype != MYOLO_F32) return MYOLO_EINVAL;
  if (!d->anchors || (!d->targets && d->nt > 0) || !d->winner || !d->ciou || !d->acc || !d->out) return MYOLO_EINVAL;
  k.nl = d->nl; k.na = d->na; k.no = d->no; k.bs = d->bs; k.nt = d->nt; k.dtype = d->dtype;
  ncell = 0;
  for (int l = 0; l < 5; ++l) {
    k.p[l] = l < d->nl ? d->p[l] : nullptr; k.gp[l] = l < d->nl ? d->gp[l] : nullptr;
    k.ny[l] = l < d->nl ? d->ny[l] : 0; k.nx[l] = l < d->nl ? d->nx[l] : 0;
    k.balance[l] = d->balance[l];
    k.cell0[l] = ncell;
    if (l < d->nl) {
      if (!d->p[l] || d->ny[l] < 1 || d->nx[l] < 1) return MYOLO_EINVAL;
      ncell += (int64_t)d->bs * d->na * d->ny[l] * d->nx[l];
    }
  }
  k.anchors = d->anchors; k.targets = d->targets;
  k.box = d->box; k.obj = d->obj; k.cls = d->cls; k.cls_pw = d->cls_pw; k.obj_pw = d->obj_pw; k.anchor_t = d->anchor_t;
  k.gr = d->gr; k.cp = d->cp; k.cn = d->cn;
  k.winner = d->winner; k.ciou = d->ciou; k.acc = d->acc; k.out = d->out; k.gp32 = d->gp32; k.gout = d->gout;
  return 0;
}

}  // namespace

extern "C" int myolo_detloss_fwd(const myolo_detloss_desc* d, void* stream) {
  DetK k; int64_t ncell;
  int r = fill_detk(d, k, ncell);
  if (r) return r;
  hipStream_t st = (hipStream_t)stream;
  const int e = myolo_fill_words2(k.winner, (size_t)ncell * sizeof(int), 0xffffffffu, k.acc, 5 * 4 * sizeof(double), 0u, st);      // -1 | 0
  if (e) return e;
  const int ncand = 5 * k.na * k.nt;
  if (ncand > 0) {
    hipLaunchKernelGGL(det_cand_fwd_kernel, dim3(grid_for(ncand, 256, 256), k.nl), dim3(256), 0, st, k);
    MYOLO_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(det_obj_fwd_kernel, dim3(grid_for(ncell, 256, 256)), dim3(256), 0, st, k, ncell);     // (one workgroup per CU: every workgroup ends with nl same-address fp64 atomics)
  hipLaunchKernelGGL(det_final_kernel, dim3(1), dim3(1), 0, st, k);
  MYOLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int myolo_detloss_bwd(const myolo_detloss_desc* d, void* stream) {
  DetK k; int64_t ncell;
  int r = fill_detk(d, k, ncell);
  if (r) return r;
  if (!k.gout) return MYOLO_EINVAL;
  for (int l = 0; l < k.nl; ++l) if (!k.gp[l]) return MYOLO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int ncand = 5 * k.na * k.nt;
  const int grid = grid_for(ncell, 256, 1024);
  if (k.dtype == MYOLO_F32) {
    float* g[5];
    for (int l = 0; l < 5; ++l) g[l] = (float*)k.gp[l];
    hipLaunchKernelGGL(det_obj_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, k, ncell, g[0], g[1], g[2], g[3], g[4]);
    if (ncand > 0)
      hipLaunchKernelGGL(det_cand_bwd_kernel, dim3(grid_for(ncand, 256, 256), k.nl), dim3(256), 0, st, k, g[0], g[1], g[2], g[3], g[4]);
  } else {
    if (!k.gp32) return MYOLO_EINVAL;     // fp32 staging: atomics on fp32, then one cast pass into the fp16 gradients
    float* g[5];
    for (int l = 0; l < 5; ++l) g[l] = k.gp32 + k.cell0[l] * k.no;
    hipLaunchKernelGGL(det_obj_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, k, ncell, g[0], g[1], g[2], g[3], g[4]);
    if (ncand > 0)
      hipLaunchKernelGGL(det_cand_bwd_kernel, dim3(grid_for(ncand, 256, 256), k.nl), dim3(256), 0, st, k, g[0], g[1], g[2], g[3], g[4]);
    for (int l = 0; l < k.nl; ++l) {
      const int64_t n = (int64_t)k.bs * k.na * k.ny[l] * k.nx[l] * k.no;
      hipLaunchKernelGGL(cast_f32_to_f16_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, st, g[l], (half_t*)k.gp[l], n);
    }
  }
  MYOLO_CHECK_LAUNCH();
  return 0;
}
