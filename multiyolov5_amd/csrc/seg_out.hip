// Output-side kernels of the segmentation / detection heads.
//   seg_upsample fwd : low-res class logits [N,h,w,C] (NHWC view, C arbitrary) -> bilinear(align_corners=True)
//                      -> strided [N,C,H,W]-logical tensor (reference models/yolo.py:163 nn.Upsample x8, detect.py:191)
//   seg_upsample bwd : transpose of the above (gather form) into the (channel-padded) low-res gradient
//   seg_argmax       : fused bilinear resize + per-pixel argmax over classes, writes only the label map
//                      (detect.py:191-193 `F.interpolate(...)` then `seg.max(axis=0)[1]`; never materialises
//                      the [19,H0,W0] logits)
//   detect_unpermute : gradient of Detect's [N,na,ny,nx,no] output back to NHWC [N,ny,nx,na*no (padded)]
//                      (autograd of yolo.py:214 view/permute)
//   detect_decode    : eval-mode box decode of the three levels into [N, sum(na*ny*nx), no] (yolo.py:216-225)
#include "myolo_dev.h"
#include <stdlib.h>

namespace {

struct Strided4 {  // generic [N,C,H,W]-logical tensor, strides in elements
  void* ptr; int64_t sn, sc, sh, sw; int dtype;
};

template <typename T> __device__ __forceinline__ float ld(const void* p, int64_t i) { return (float)((const T*)p)[i]; }
__device__ __forceinline__ float ld_any(const void* p, int64_t i, int dt) {
  return dt == MYOLO_F16 ? (float)((const half_t*)p)[i] : ((const float*)p)[i];
}
__device__ __forceinline__ void st_any(void* p, int64_t i, int dt, float v) {
  if (dt == MYOLO_F16) ((half_t*)p)[i] = (half_t)v; else ((float*)p)[i] = v;
}

constexpr int MAXC = 32;   // classes handled per pixel in registers (n_segcls = 19)

__global__ __launch_bounds__(256) void seg_up_fwd_kernel(myolo_tensor low, Strided4 out, int H, int W, float sy, float sx) {
  const int C = low.c;
  const int64_t total = (int64_t)low.n * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W); const int y = (int)((i / W) % H); const int n = (int)(i / ((int64_t)W * H));
    const float fy = sy * (float)y, fx = sx * (float)x;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + 1 < low.h ? y0 + 1 : low.h - 1, x1 = x0 + 1 < low.w ? x0 + 1 : low.w - 1;
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const int64_t b00 = (int64_t)n * low.sn + (int64_t)y0 * low.sh + (int64_t)x0 * low.sw;
    const int64_t b01 = (int64_t)n * low.sn + (int64_t)y0 * low.sh + (int64_t)x1 * low.sw;
    const int64_t b10 = (int64_t)n * low.sn + (int64_t)y1 * low.sh + (int64_t)x0 * low.sw;
    const int64_t b11 = (int64_t)n * low.sn + (int64_t)y1 * low.sh + (int64_t)x1 * low.sw;
    const int64_t ob = (int64_t)n * out.sn + (int64_t)y * out.sh + (int64_t)x * out.sw;
    for (int c = 0; c < C; ++c) {
      const float a = ld_any(low.ptr, b00 + c, low.dtype), b = ld_any(low.ptr, b01 + c, low.dtype);
      const float cc = ld_any(low.ptr, b10 + c, low.dtype), d = ld_any(low.ptr, b11 + c, low.dtype);
      const float v = (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * cc + lx * d);
      st_any(out.ptr, ob + (int64_t)c * out.sc, out.dtype, v);
    }
  }
}

__global__ __launch_bounds__(256) void seg_argmax_kernel(myolo_tensor low, void* labels, int label_dtype, int H, int W,
                                                         float sy, float sx) {
  const int C = low.c;
  const int64_t total = (int64_t)low.n * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W); const int y = (int)((i / W) % H); const int n = (int)(i / ((int64_t)W * H));
    const float fy = sy * (float)y, fx = sx * (float)x;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + 1 < low.h ? y0 + 1 : low.h - 1, x1 = x0 + 1 < low.w ? x0 + 1 : low.w - 1;
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const int64_t b00 = (int64_t)n * low.sn + (int64_t)y0 * low.sh + (int64_t)x0 * low.sw;
    const int64_t b01 = (int64_t)n * low.sn + (int64_t)y0 * low.sh + (int64_t)x1 * low.sw;
    const int64_t b10 = (int64_t)n * low.sn + (int64_t)y1 * low.sh + (int64_t)x0 * low.sw;
    const int64_t b11 = (int64_t)n * low.sn + (int64_t)y1 * low.sh + (int64_t)x1 * low.sw;
    float best = -INFINITY; int arg = 0;
    for (int c = 0; c < C; ++c) {
      const float a = ld_any(low.ptr, b00 + c, low.dtype), b = ld_any(low.ptr, b01 + c, low.dtype);
      const float cc = ld_any(low.ptr, b10 + c, low.dtype), d = ld_any(low.ptr, b11 + c, low.dtype);
      float v = (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * cc + lx * d);
      if (low.dtype == MYOLO_F16) v = (float)(half_t)v;     // the reference materialises fp16 logits before max
      if (v > best) { best = v; arg = c; }                   // first maximum wins (torch.max)
    }
    if (label_dtype == MYOLO_I64) ((int64_t*)labels)[i] = arg; else ((uint8_t*)labels)[i] = (uint8_t)arg;
  }
}

// tiled form of seg_argmax_kernel: a workgroup owns 256 consecutive pixels of one output row; the two low-resolution rows it interpolates
// between go through LDS ONCE (x8 upsample: ~35 columns x C classes x 2 rows as fp32), every thread then reads its 4 taps x C classes from
// there -- the per-pixel kernel above issued 4 x C two-byte global loads per pixel (79 us for a 2048x1024 label map; the map is 16 MB of
// int64 writes = ~4 us of HBM time).  Same expression per class, same first-maximum rule.
template <typename T>
__global__ __launch_bounds__(256) void seg_argmax_tile_kernel(myolo_tensor low, void* labels, int label_dtype, int H, int W, float sy,
                                                              float sx, int tiles_x, int maxcol) {
  extern __shared__ float sl[];                                   // [2][maxcol][C]
  const int C = low.c;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int y = bid % H; const int n = bid / H;
  const int xa = tx * 256, xb = (xa + 255 < W ? xa + 255 : W - 1);
  const float fy = sy * (float)y;
  const int y0 = (int)fy;
  const int y1 = y0 + 1 < low.h ? y0 + 1 : low.h - 1;
  const float ly = fy - (float)y0;
  const int c0 = (int)(sx * (float)xa);
  int c1 = (int)(sx * (float)xb) + 1;
  if (c1 > low.w - 1) c1 = low.w - 1;
  const int ncol = c1 - c0 + 1;                                   // <= maxcol (host)
  const T* src = reinterpret_cast<const T*>(low.ptr);
  for (int e = threadIdx.x; e < 2 * ncol * C; e += 256) {
    const int r = e / (ncol * C); const int rem = e - r * ncol * C;
    const int col = rem / C, c = rem - col * C;
    sl[(r * maxcol + col) * C + c] = (float)src[(int64_t)n * low.sn + (int64_t)(r ? y1 : y0) * low.sh + (int64_t)(c0 + col) * low.sw + c];
  }
  __syncthreads();
  const int x = xa + threadIdx.x;
  if (x >= W) return;
  const float fx = sx * (float)x;
  const int x0 = (int)fx;
  const int x1 = x0 + 1 < low.w ? x0 + 1 : low.w - 1;
  const float lx = fx - (float)x0;
  const float* p00 = sl + (x0 - c0) * C; const float* p01 = sl + (x1 - c0) * C;
  const float* p10 = sl + (maxcol + x0 - c0) * C; const float* p11 = sl + (maxcol + x1 - c0) * C;
  float best = -INFINITY; int arg = 0;
  for (int c = 0; c < C; ++c) {
    const float a = p00[c], b = p01[c], cc = p10[c], d = p11[c];
    float v = (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * cc + lx * d);
    if (sizeof(T) == 2) v = (float)(half_t)v;                     // the reference materialises fp16 logits before max
    if (v > best) { best = v; arg = c; }                          // first maximum wins (torch.max)
  }
  const int64_t i = ((int64_t)n * H + y) * W + x;
  if (label_dtype == MYOLO_I64) ((int64_t*)labels)[i] = arg; else ((uint8_t*)labels)[i] = (uint8_t)arg;
}

__device__ __forceinline__ void out_range(int i, int in, int out, float s, int& lo, int& hi) {
  if (out == 1 || in == 1 || s <= 0.f) { lo = 0; hi = out - 1; return; }
  lo = (int)floorf(((float)i - 1.f) / s) - 1; hi = (int)ceilf(((float)i + 1.f) / s) + 1;
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}

// one thread per (n, iy, ix, c) of the low-res gradient
__global__ __launch_bounds__(256) void seg_up_bwd_kernel(Strided4 g, int H, int W, myolo_tensor glow, float sy, float sx,
                                                         int acc, const float* scale) {
  const int C = glow.c;
  const float gs = scale ? scale[0] : 1.f;
  const int64_t total = (int64_t)glow.n * glow.h * glow.w * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    const int c = (int)(r % C); r /= C;
    const int ix = (int)(r % glow.w); r /= glow.w;
    const int iy = (int)(r % glow.h); const int n = (int)(r / glow.h);
    int ylo, yhi, xlo, xhi;
    out_range(iy, glow.h, H, sy, ylo, yhi);
    out_range(ix, glow.w, W, sx, xlo, xhi);
    float a = 0.f;
    for (int oy = ylo; oy <= yhi; ++oy) {
      const float fy = sy * (float)oy; const int y0 = (int)fy; const int y1 = y0 + 1 < glow.h ? y0 + 1 : glow.h - 1;
      const float ly = fy - (float)y0;
      float wy = 0.f;
      if (y0 == iy) wy += 1.f - ly;
      if (y1 == iy) wy += ly;
      if (wy == 0.f) continue;
      for (int ox = xlo; ox <= xhi; ++ox) {
        const float fx = sx * (float)ox; const int x0 = (int)fx; const int x1 = x0 + 1 < glow.w ? x0 + 1 : glow.w - 1;
        const float lx = fx - (float)x0;
        float wx = 0.f;
        if (x0 == ix) wx += 1.f - lx;
        if (x1 == ix) wx += lx;
        if (wx == 0.f) continue;
        a += wy * wx * ld_any(g.ptr, (int64_t)n * g.sn + (int64_t)c * g.sc + (int64_t)oy * g.sh + (int64_t)ox * g.sw, g.dtype);
      }
    }
    const int64_t o = (int64_t)n * glow.sn + (int64_t)iy * glow.sh + (int64_t)ix * glow.sw + c;
    a *= gs;
    if (acc) a += ld_any(glow.ptr, o, glow.dtype);
    st_any(glow.ptr, o, glow.dtype, a);
  }
}


// ---- fast paths for dense channels-last storage ([N,H,W,C] contiguous: the layout Model.forward hands out) -------------
constexpr int STRIP = 256;
template <typename T>
__global__ __launch_bounds__(STRIP) void seg_up_fwd_cl_kernel(myolo_tensor low, T* out, int H, int W, float sy, float sx) {
  __shared__ __attribute__((aligned(16))) T buf[STRIP * MAXC];
  const int C = low.c;
  const int strips = (W + STRIP - 1) / STRIP;
  int b = blockIdx.x;
  const int xs = b % strips; b /= strips;
  const int y = b % H; const int n = b / H;
  const int x0 = xs * STRIP;
  const int npix = W - x0 < STRIP ? W - x0 : STRIP;
  const int x = x0 + threadIdx.x;
  if ((int)threadIdx.x < npix) {
    const float fy = sy * (float)y, fx = sx * (float)x;
    const int y0 = (int)fy, xx0 = (int)fx;
    const int y1 = y0 + 1 < low.h ? y0 + 1 : low.h - 1, x1 = xx0 + 1 < low.w ? xx0 + 1 : low.w - 1;
    const float ly = fy - (float)y0, lx = fx - (float)xx0;
    const T* lp = reinterpret_cast<const T*>(low.ptr) + (int64_t)n * low.sn;
    const T* p00 = lp + (int64_t)y0 * low.sh + (int64_t)xx0 * low.sw;
    const T* p01 = lp + (int64_t)y0 * low.sh + (int64_t)x1 * low.sw;
    const T* p10 = lp + (int64_t)y1 * low.sh + (int64_t)xx0 * low.sw;
    const T* p11 = lp + (int64_t)y1 * low.sh + (int64_t)x1 * low.sw;
    for (int c = 0; c < C; ++c) {
      const float a = (float)p00[c], bb = (float)p01[c], cc = (float)p10[c], d = (float)p11[c];
      buf[threadIdx.x * C + c] = (T)((1.f - ly) * ((1.f - lx) * a + lx * bb) + ly * ((1.f - lx) * cc + lx * d));
    }
  }
  __syncthreads();
  strip_store(out + (((int64_t)n * H + y) * W + x0) * C, buf, npix * C);
}

// transpose of the above for RLOW low-res rows x one segment of LXP low-res pixels.  Separable: (1) every thread owns up
// to 24 elements of the hi-res strip and streams the rows of the footprint straight from HBM into fp32 registers,
// weighting each row into the (at most two) low rows it touches -- no LDS, no barrier, next row in flight; (2) the
// column sums go to LDS once and each (low x, class) pair folds its x footprint.
constexpr int LXP = 32;
constexpr int RLOW = 2;
constexpr int UPB_ELEMS = 24;                    // strip elements per thread
template <typename T>
__global__ __launch_bounds__(256) void seg_up_bwd_cl_kernel(const T* g, int H, int W, myolo_tensor glow, float sy, float sx,
                                                            int acc, int xalign, int lds_stride, const float* scale) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* fbuf = reinterpret_cast<float*>(smem_raw);            // [RLOW][lds_stride]
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int PRE = UPB_ELEMS / VEC;
  const int C = glow.c;
  const int segs = (glow.w + LXP - 1) / LXP;
  const int rgroups = (glow.h + RLOW - 1) / RLOW;
  int b = blockIdx.x;
  const int xsg = b % segs; b /= segs;
  const int iy0 = (b % rgroups) * RLOW; const int n = b / rgroups;
  const int nr = glow.h - iy0 < RLOW ? glow.h - iy0 : RLOW;
  const int lx0 = xsg * LXP;
  const int nlx = glow.w - lx0 < LXP ? glow.w - lx0 : LXP;
  int ylo, yhi, xlo, xhi, t0, t1;
  out_range(iy0, glow.h, H, sy, ylo, t0);
  out_range(iy0 + nr - 1, glow.h, H, sy, t1, yhi);
  out_range(lx0, glow.w, W, sx, xlo, t0);
  out_range(lx0 + nlx - 1, glow.w, W, sx, t1, xhi);
  xlo = xlo / xalign * xalign;
  const int npix = xhi - xlo + 1;
  const int nvec = (npix * C + VEC - 1) / VEC;               // rows are whole vectors, so the round-up stays inside the row
  float av[RLOW][UPB_ELEMS];
#pragma unroll
  for (int r = 0; r < RLOW; ++r)
#pragma unroll
    for (int e = 0; e < UPB_ELEMS; ++e) av[r][e] = 0.f;
  const int64_t rowstride = (int64_t)W * C;
  const T* base = g + ((int64_t)n * H * W + xlo) * C;
  int vidx[PRE];
#pragma unroll
  for (int k = 0; k < PRE; ++k) { const int v = threadIdx.x + k * 256; vidx[k] = v < nvec ? v : nvec - 1; }   // clamped: loads stay unconditional
  uint4 cur[PRE], nxt[PRE];
#pragma unroll
  for (int k = 0; k < PRE; ++k) cur[k] = ldg16(reinterpret_cast<const uint4*>(base + (int64_t)ylo * rowstride) + vidx[k]);
  for (int oy = ylo; oy <= yhi; ++oy) {
    const int on = oy + 1 <= yhi ? oy + 1 : yhi;
#pragma unroll
    for (int k = 0; k < PRE; ++k) nxt[k] = ldg16(reinterpret_cast<const uint4*>(base + (int64_t)on * rowstride) + vidx[k]);
    const float fy = sy * (float)oy; const int y0 = (int)fy; const int y1 = y0 + 1 < glow.h ? y0 + 1 : glow.h - 1;
    const float ly = fy - (float)y0;
#pragma unroll
    for (int r = 0; r < RLOW; ++r) {
      float wy = 0.f;
      if (y0 == iy0 + r) wy += 1.f - ly;
      if (y1 == iy0 + r) wy += ly;
      if (wy != 0.f) {                              // uniform over the workgroup
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
          const T* e = reinterpret_cast<const T*>(&cur[k]);
#pragma unroll
          for (int j = 0; j < VEC; ++j) av[r][k * VEC + j] += wy * (float)e[j];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < PRE; ++k) cur[k] = nxt[k];
  }
#pragma unroll
  for (int r = 0; r < RLOW; ++r)
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      const int v = threadIdx.x + k * 256;
      if (v < nvec) {
        float4* d = reinterpret_cast<float4*>(fbuf + (size_t)r * lds_stride + (size_t)v * VEC);
#pragma unroll
        for (int j = 0; j < VEC / 4; ++j)
          d[j] = float4{av[r][k * VEC + j * 4], av[r][k * VEC + j * 4 + 1], av[r][k * VEC + j * 4 + 2], av[r][k * VEC + j * 4 + 3]};
      }
    }
  __syncthreads();
  const float gs = scale ? scale[0] : 1.f;
  const int npairs = nlx * C;
  for (int p = threadIdx.x; p < npairs; p += 256) {
    const int li = p / C, c = p - li * C;
    const int ix = lx0 + li;
    int pxlo, pxhi;
    out_range(ix, glow.w, W, sx, pxlo, pxhi);
    float s[RLOW];
#pragma unroll
    for (int r = 0; r < RLOW; ++r) s[r] = 0.f;
    const float* bp = fbuf + (pxlo - xlo) * C + c;
    for (int ox = pxlo; ox <= pxhi; ++ox, bp += C) {
      const float fx = sx * (float)ox; const int x0 = (int)fx; const int x1 = x0 + 1 < glow.w ? x0 + 1 : glow.w - 1;
      const float lx = fx - (float)x0;
      float wx = 0.f;
      if (x0 == ix) wx += 1.f - lx;
      if (x1 == ix) wx += lx;
#pragma unroll
      for (int r = 0; r < RLOW; ++r) s[r] += wx * bp[(size_t)r * lds_stride];
    }
#pragma unroll
    for (int r = 0; r < RLOW; ++r) {
      if (r >= nr) break;
      T* o = reinterpret_cast<T*>(glow.ptr) + (int64_t)n * glow.sn + (int64_t)(iy0 + r) * glow.sh + (int64_t)ix * glow.sw + c;
      float v = s[r] * gs;
      if (acc) v += (float)*o;
      *o = (T)v;
    }
  }
}

inline bool dense_cl(const void* p, int dt, int C, int H, int W, int64_t sn, int64_t sc, int64_t sh, int64_t sw) {
  const int es = dt == MYOLO_F16 ? 2 : 4;
  return (dt == MYOLO_F16 || dt == MYOLO_F32) && sc == 1 && sw == C && sh == (int64_t)W * C && sn == (int64_t)H * W * C &&
         ((int64_t)W * C * es) % 16 == 0 && ((uintptr_t)p & 15) == 0 && C <= MAXC;
}


// ---- evaluation counters (reference utils/metrics.py:234-275, called per batch by test.py:31-65 seg_validation) ---------
// counts[0] = pixel_correct, counts[1] = pixel_labeled, then area_inter[n], area_pred[n], area_lab[n] (all over target >= 0;
// np.histogram(x, bins=n, range=(1, n)) of the 1-based labels is the identity binning).  Bit-exact integer work.
__global__ __launch_bounds__(256) void seg_metrics_kernel(const void* pred, int pdt, const int64_t* tgt, int64_t total, int ncls,
                                                          unsigned long long* counts) {
  __shared__ unsigned int h[3 * MAXC + 2];
  for (int i = threadIdx.x; i < 3 * MAXC + 2; i += 256) h[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = tgt[i];
    if (t < 0) continue;                                            // target + 1 > 0 (metrics.py:241,243 / 265)
    const int pr = pdt == MYOLO_I64 ? (int)((const int64_t*)pred)[i] : (int)((const uint8_t*)pred)[i];
    atomicAdd(&h[1], 1u);
    if (pr == (int)t) { atomicAdd(&h[0], 1u); if (pr < ncls) atomicAdd(&h[2 + pr], 1u); }
    if (pr >= 0 && pr < ncls) atomicAdd(&h[2 + MAXC + pr], 1u);
    if (t < ncls) atomicAdd(&h[2 + 2 * MAXC + (int)t], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 + 3 * ncls; i += 256) {
    const int src = i < 2 ? i : 2 + ((i - 2) / ncls) * MAXC + (i - 2) % ncls;
    if (h[src]) atomicAdd(counts + i, (unsigned long long)h[src]);
  }
}

// g: dense [N,na,ny,nx,no] (dtype gdt) -> out NHWC view [N,ny,nx,na*no] (padded channels untouched)
__global__ __launch_bounds__(256) void detect_unpermute_kernel(const void* g, int gdt, int na, int no, myolo_tensor out) {
  const int C = na * no;
  const int64_t total = (int64_t)out.n * out.h * out.w * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    const int c = (int)(r % C); r /= C;
    const int x = (int)(r % out.w); r /= out.w;
    const int y = (int)(r % out.h); const int n = (int)(r / out.h);
    const int a = c / no, o = c - a * no;
    const float v = ld_any(g, ((((int64_t)n * na + a) * out.h + y) * out.w + x) * no + o, gdt);
    st_any(out.ptr, (int64_t)n * out.sn + (int64_t)y * out.sh + (int64_t)x * out.sw + c, out.dtype, v);
  }
}

// raw: dense [N,na,ny,nx,no]; z: dense [N, A_total, no], rows [row0, row0+na*ny*nx) of each image, order (a,y,x)
__global__ __launch_bounds__(256) void detect_decode_kernel(const void* raw, int dt, int N, int na, int ny, int nx, int no,
                                                            float stride, float aw0, float ah0, float aw1, float ah1,
                                                            float aw2, float ah2, void* z, int64_t a_total, int64_t row0) {
  const int64_t per_img = (int64_t)na * ny * nx;
  const int64_t total = (int64_t)N * per_img;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / per_img); const int64_t r = i - (int64_t)n * per_img;
    const int a = (int)(r / ((int64_t)ny * nx)); const int rem = (int)(r - (int64_t)a * ny * nx);
    const int y = rem / nx, x = rem - y * nx;
    const float aw = a == 0 ? aw0 : (a == 1 ? aw1 : aw2), ah = a == 0 ? ah0 : (a == 1 ? ah1 : ah2);
    const int64_t src = i * no;
    const int64_t dst = ((int64_t)n * a_total + row0 + r) * no;
    for (int o = 0; o < no; ++o) {
      float s = sigmoid_f(ld_any(raw, src + o, dt));
      if (dt == MYOLO_F16) s = (float)(half_t)s;          // reference: y = x.sigmoid() materialised in fp16
      float v;
      if (o == 0) v = (s * 2.f - 0.5f + (float)x) * stride;
      else if (o == 1) v = (s * 2.f - 0.5f + (float)y) * stride;
      else if (o == 2) v = (s * 2.f) * (s * 2.f) * aw;
      else if (o == 3) v = (s * 2.f) * (s * 2.f) * ah;
      else v = s;
      st_any(z, dst + o, dt, v);
    }
  }
}

}  // namespace

extern "C" int myolo_seg_upsample_fwd(const myolo_tensor* low, void* out, int out_dtype, int H, int W, int64_t sn,
                                      int64_t sc, int64_t sh, int64_t sw, void* stream) {
  if (!low || !low->ptr || !out || H < 1 || W < 1) return MYOLO_EINVAL;
  Strided4 o{out, sn, sc, sh, sw, out_dtype};
  const float sy = H > 1 ? (float)(low->h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(low->w - 1) / (float)(W - 1) : 0.f;
  if (out_dtype == low->dtype && dense_cl(out, out_dtype, low->c, H, W, sn, sc, sh, sw) && (H > low->h || W > low->w)) {
    const int64_t blocks = (int64_t)low->n * H * ((W + STRIP - 1) / STRIP);
    if (out_dtype == MYOLO_F16)
      hipLaunchKernelGGL(seg_up_fwd_cl_kernel<half_t>, dim3((int)blocks), dim3(STRIP), 0, (hipStream_t)stream, *low, (half_t*)out, H, W, sy, sx);
    else
      hipLaunchKernelGGL(seg_up_fwd_cl_kernel<float>, dim3((int)blocks), dim3(STRIP), 0, (hipStream_t)stream, *low, (float*)out, H, W, sy, sx);
    MYOLO_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(seg_up_fwd_kernel, dim3(grid_for((int64_t)low->n * H * W, 256, 8192)), dim3(256), 0,
                     (hipStream_t)stream, *low, o, H, W, sy, sx);
  MYOLO_CHECK_LAUNCH();
  return 0;
}
extern "C" int myolo_seg_upsample_bwd(const void* g, int g_dtype, int H, int W, int64_t sn, int64_t sc, int64_t sh,
                                      int64_t sw, const myolo_tensor* glow, int accumulate, const float* scale, void* stream) {
  if (!glow || !glow->ptr || !g) return MYOLO_EINVAL;
  Strided4 gg{const_cast<void*>(g), sn, sc, sh, sw, g_dtype};
  const float sy = H > 1 ? (float)(glow->h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(glow->w - 1) / (float)(W - 1) : 0.f;
  if (g_dtype == glow->dtype && dense_cl(g, g_dtype, glow->c, H, W, sn, sc, sh, sw) && H >= 2 * glow->h && W >= 2 * glow->w &&
      W <= 8 * glow->w) {
    const int es = g_dtype == MYOLO_F16 ? 2 : 4;
    int gcd = 16, t = glow->c * es;
    while (t) { const int r = gcd % t; gcd = t; t = r; }
    const int xalign = 16 / gcd;                                   // pixels per 16-byte boundary of the row
    const int scale_x = (W + glow->w - 1) / glow->w;
    const int maxpix = (LXP + 3) * scale_x + xalign + 8;
    const int vec = 16 / es;
    const int lds_stride = (maxpix * glow->c + vec - 1) / vec * vec;
    const int smem = RLOW * lds_stride * (int)sizeof(float);
    const int64_t blocks = (int64_t)glow->n * ((glow->h + RLOW - 1) / RLOW) * ((glow->w + LXP - 1) / LXP);
    if (smem <= 64 * 1024 && lds_stride <= 256 * UPB_ELEMS) {
      if (g_dtype == MYOLO_F16)
        hipLaunchKernelGGL(seg_up_bwd_cl_kernel<half_t>, dim3((int)blocks), dim3(256), smem, (hipStream_t)stream, (const half_t*)g, H, W, *glow, sy, sx, accumulate, xalign, lds_stride, scale);
      else
        hipLaunchKernelGGL(seg_up_bwd_cl_kernel<float>, dim3((int)blocks), dim3(256), smem, (hipStream_t)stream, (const float*)g, H, W, *glow, sy, sx, accumulate, xalign, lds_stride, scale);
      MYOLO_CHECK_LAUNCH();
      return 0;
    }
  }
  hipLaunchKernelGGL(seg_up_bwd_kernel, dim3(grid_for((int64_t)glow->n * glow->h * glow->w * glow->c, 256, 8192)), dim3(256),
                     0, (hipStream_t)stream, gg, H, W, *glow, sy, sx, accumulate, scale);
  MYOLO_CHECK_LAUNCH();
  return 0;
}
extern "C" int myolo_seg_argmax(const myolo_tensor* low, void* labels, int label_dtype, int H, int W, void* stream) {
  if (!low || !low->ptr || !labels || (label_dtype != MYOLO_U8 && label_dtype != MYOLO_I64)) return MYOLO_EINVAL;
  const float sy = H > 1 ? (float)(low->h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(low->w - 1) / (float)(W - 1) : 0.f;
  {
    // up-scaling (detect.py:191: the x8 head output, or a resize to the source frame): rows staged through LDS
    const int maxcol = (int)(sx * 255.f) + 3;
    const int smem = 2 * maxcol * low->c * 4;
    const int tiles_x = (W + 255) / 256;
    const int64_t blocks = (int64_t)low->n * H * tiles_x;
    if ((low->dtype == MYOLO_F16 || low->dtype == MYOLO_F32) && low->c <= 32 && smem <= 48 * 1024 && blocks < 0x7fffffff && W >= 64) {
      if (low->dtype == MYOLO_F16)
        hipLaunchKernelGGL(seg_argmax_tile_kernel<half_t>, dim3((int)blocks), dim3(256), smem, (hipStream_t)stream, *low, labels, label_dtype, H,
                           W, sy, sx, tiles_x, maxcol);
      else
        hipLaunchKernelGGL(seg_argmax_tile_kernel<float>, dim3((int)blocks), dim3(256), smem, (hipStream_t)stream, *low, labels, label_dtype, H,
                           W, sy, sx, tiles_x, maxcol);
      MYOLO_CHECK_LAUNCH();
      return 0;
    }
  }
  hipLaunchKernelGGL(seg_argmax_kernel, dim3(grid_for((int64_t)low->n * H * W, 256, 8192)), dim3(256), 0,
                     (hipStream_t)stream, *low, labels, label_dtype, H, W, sy, sx);
  MYOLO_CHECK_LAUNCH();
  return 0;
}
extern "C" int myolo_detect_unpermute(const void* g, int g_dtype, int na, int no, const myolo_tensor* out, void* stream) {
  if (!g || !out || !out->ptr || out->c < na * no) return MYOLO_EINVAL;
  hipLaunchKernelGGL(detect_unpermute_kernel, dim3(grid_for((int64_t)out->n * out->h * out->w * na * no, 256)), dim3(256),
                     0, (hipStream_t)stream, g, g_dtype, na, no, *out);
  MYOLO_CHECK_LAUNCH();
  return 0;
}
extern "C" int myolo_detect_decode(const void* raw, int dtype, int n, int na, int ny, int nx, int no, float stride,
                                   const float* anchor_wh_px /* host, na*2 */, void* z, int64_t a_total, int64_t row0,
                                   void* stream) {
  if (!raw || !z || !anchor_wh_px || na != 3) return MYOLO_EINVAL;
  hipLaunchKernelGGL(detect_decode_kernel, dim3(grid_for((int64_t)n * na * ny * nx, 256)), dim3(256), 0,
                     (hipStream_t)stream, raw, dtype, n, na, ny, nx, no, stride, anchor_wh_px[0], anchor_wh_px[1],
                     anchor_wh_px[2], anchor_wh_px[3], anchor_wh_px[4], anchor_wh_px[5], z, a_total, row0);
  MYOLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int myolo_seg_metrics(const void* pred, int pred_dtype, const int64_t* target, int64_t total, int nclass,
                                 uint64_t* counts, void* stream) {
  if (!pred || !target || !counts || total < 1 || nclass < 1 || nclass > MAXC || (pred_dtype != MYOLO_U8 && pred_dtype != MYOLO_I64))
    return MYOLO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(counts, 0, (2 + 3 * nclass) * sizeof(uint64_t), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(seg_metrics_kernel, dim3(grid_for(total, 256, 1024)), dim3(256), 0, st, pred, pred_dtype, target, total, nclass,
                     reinterpret_cast<unsigned long long*>(counts));
  MYOLO_CHECK_LAUNCH();
  return 0;
}
