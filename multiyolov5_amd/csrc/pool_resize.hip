// Pooling / resampling / glue kernels of the hot path (all HBM- or L2-bound, NHWC 16-byte vectors, fp32 math).
//   SPP max pools k=5,9,13 stride 1 (reference models/common.py:170)        fwd (+first-max index) / bwd (gather)
//   nearest 2x upsample or plain copy into a concat slice (yaml:31,36; common.py:589)
//   bilinear align_corners=True resize (models/yolo.py:57..174, common.py:534-537)  fwd / bwd (gather form, no atomics)
//   AdaptiveAvgPool2d(k) (common.py:521-524,214)                               fwd / bwd
//   FFM gate feat*att+feat (common.py:228-229)                                 fwd / bwd
#include "myolo_dev.h"
#include <string.h>

// myolo_set_option("spp_naive", 1): the per-output-vector SPP kernels (what planes too large for the LDS take) for every map (tests)
static int g_spp_naive = 0;
static int g_aap_wgs = 256;        // myolo_set_option("pool_aap_wgs", n): workgroup target of the one-pass pyramid pools (myolo_adaptive_avgpool_fwd_multi)
static int g_spp_bwd_form = 0;     // myolo_set_option("spp_bwd_form", 1): round 5's plane kernel (a thread = a pixel x 8 channels) instead of round 6's channel-lane kernel
int myolo_pool_set(const char* name, int value) {
  if (!strcmp(name, "spp_naive")) { g_spp_naive = value; return 0; }
  if (!strcmp(name, "spp_bwd_form")) { g_spp_bwd_form = value; return 0; }
  if (!strcmp(name, "pool_aap_wgs")) { if (value < 1) return MYOLO_EINVAL; g_aap_wgs = value; return 0; }
  return MYOLO_EINVAL;
}
#include <stdlib.h>

namespace {

inline bool vec_ok(const myolo_tensor* t) {
  const int seg = t->dtype == MYOLO_F16 ? 8 : 4;
  return t && t->ptr && (t->dtype == MYOLO_F16 || t->dtype == MYOLO_F32) && t->c % seg == 0 && t->sw % seg == 0 &&
         t->sh % seg == 0 && t->sn % seg == 0 && ((uintptr_t)t->ptr & 15) == 0;
}

// decode vector index -> (n, y, x, cg) for a tensor with G channel groups
__device__ __forceinline__ void dec(int64_t v, int G, int W, int H, int& n, int& y, int& x, int& cg) {
  if ((uint64_t)v < 0x80000000ull) {            // the usual case: three 32-bit divisions instead of three emulated 64-bit ones
    uint32_t u = (uint32_t)v, q = u / (uint32_t)G;
    cg = (int)(u - q * (uint32_t)G); u = q; q = u / (uint32_t)W;
    x = (int)(u - q * (uint32_t)W); u = q; q = u / (uint32_t)H;
    y = (int)(u - q * (uint32_t)H);
    n = (int)q;
    return;
  }
  cg = (int)(v % G); v /= G;
  x = (int)(v % W); v /= W;
  y = (int)(v % H);
  n = (int)(v / H);
}

#define GRID_STRIDE(v, total) \
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < (total); v += (int64_t)gridDim.x * blockDim.x)

// ---------------------------------------------------------------- SPP max pools
template <typename T, bool IDX>
__global__ __launch_bounds__(256) void spp_fwd_kernel(myolo_tensor x, myolo_tensor o5, myolo_tensor o9, myolo_tensor o13,
                                                      uint8_t* idx) {
  constexpr int SEG = ET<T>::SEG;
  const int G = x.c / SEG;
  const int64_t total = (int64_t)x.n * x.h * x.w * G;
  const int64_t plane = (int64_t)x.n * x.h * x.w * x.c;   // idx plane size per pool
  GRID_STRIDE(v, total) {
    int n, y, xx, cg;
    dec(v, G, x.w, x.h, n, y, xx, cg);
    float m5[SEG], m9[SEG], m13[SEG];
    int i5[SEG], i9[SEG], i13[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) { m5[i] = m9[i] = m13[i] = -INFINITY; i5[i] = i9[i] = i13[i] = 0; }
    // row-major scan of the 13x13 window; strict '>' keeps the first maximum (ATen max_pool2d semantics)
    for (int dy = -6; dy <= 6; ++dy) {
      const int iy = y + dy;
      if (iy < 0 || iy >= x.h) continue;
      for (int dx = -6; dx <= 6; ++dx) {
        const int ix = xx + dx;
        if (ix < 0 || ix >= x.w) continue;
        float f[SEG];
        Vec<T>::unpack(ldg16(vptr<T>(x, n, iy, ix) + cg * SEG), f);
        const bool in9 = dy >= -4 && dy <= 4 && dx >= -4 && dx <= 4;
        const bool in5 = dy >= -2 && dy <= 2 && dx >= -2 && dx <= 2;
#pragma unroll
        for (int i = 0; i < SEG; ++i) {
          if (f[i] > m13[i]) { m13[i] = f[i]; i13[i] = (dy + 6) * 13 + dx + 6; }
          if (in9 && f[i] > m9[i]) { m9[i] = f[i]; i9[i] = (dy + 4) * 9 + dx + 4; }
          if (in5 && f[i] > m5[i]) { m5[i] = f[i]; i5[i] = (dy + 2) * 5 + dx + 2; }
        }
      }
    }
    stg16(vptr<T>(o5, n, y, xx) + cg * SEG, Vec<T>::pack(m5));
    stg16(vptr<T>(o9, n, y, xx) + cg * SEG, Vec<T>::pack(m9));
    stg16(vptr<T>(o13, n, y, xx) + cg * SEG, Vec<T>::pack(m13));
    if (IDX) {
      const int64_t e = (((int64_t)n * x.h + y) * x.w + xx) * x.c + cg * SEG;
#pragma unroll
      for (int i = 0; i < SEG; ++i) {
        idx[e + i] = (uint8_t)i5[i];
        idx[plane + e + i] = (uint8_t)i9[i];
        idx[2 * plane + e + i] = (uint8_t)i13[i];
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void spp_bwd_kernel(myolo_tensor g5, myolo_tensor g9, myolo_tensor g13,
                                                      const uint8_t* __restrict__ idx, myolo_tensor gx, int acc) {
  constexpr int SEG = ET<T>::SEG;
  const int G = gx.c / SEG;
  const int64_t total = (int64_t)gx.n * gx.h * gx.w * G;
  const int64_t plane = (int64_t)gx.n * gx.h * gx.w * gx.c;
  GRID_STRIDE(v, total) {
    int n, y, xx, cg;
    dec(v, G, gx.w, gx.h, n, y, xx, cg);
    float a[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) a[i] = 0.f;
    // output (oy,ox) selected input (y,xx) iff its recorded window offset equals (y-oy+r, xx-ox+r)
    for (int dy = -6; dy <= 6; ++dy) {
      const int oy = y + dy;
      if (oy < 0 || oy >= gx.h) continue;
      for (int dx = -6; dx <= 6; ++dx) {
        const int ox = xx + dx;
        if (ox < 0 || ox >= gx.w) continue;
        const int64_t e = (((int64_t)n * gx.h + oy) * gx.w + ox) * gx.c + cg * SEG;
        {
          float f[SEG];
          Vec<T>::unpack(ldg16(vptr<T>(g13, n, oy, ox) + cg * SEG), f);
          const int want = (6 - dy) * 13 + (6 - dx);
#pragma unroll
          for (int i = 0; i < SEG; ++i) if (idx[2 * plane + e + i] == want) a[i] += f[i];
        }
        if (dy >= -4 && dy <= 4 && dx >= -4 && dx <= 4) {
          float f[SEG];
          Vec<T>::unpack(ldg16(vptr<T>(g9, n, oy, ox) + cg * SEG), f);
          const int want = (4 - dy) * 9 + (4 - dx);
#pragma unroll
          for (int i = 0; i < SEG; ++i) if (idx[plane + e + i] == want) a[i] += f[i];
        }
        if (dy >= -2 && dy <= 2 && dx >= -2 && dx <= 2) {
          float f[SEG];
          Vec<T>::unpack(ldg16(vptr<T>(g5, n, oy, ox) + cg * SEG), f);
          const int want = (2 - dy) * 5 + (2 - dx);
#pragma unroll
          for (int i = 0; i < SEG; ++i) if (idx[e + i] == want) a[i] += f[i];
        }
      }
    }
    T* gp = vptr<T>(gx, n, y, xx) + cg * SEG;
    if (acc) {
      float o[SEG];
      Vec<T>::unpack(ldg16(gp), o);
#pragma unroll
      for (int i = 0; i < SEG; ++i) a[i] += o[i];
    }
    stg16(gp, Vec<T>::pack(a));
  }
}

// SPP on maps whose plane fits in LDS (the stride-32 map of a 512x1024 image is 16x32): one workgroup owns the whole plane of one
// (image, channel group) in LDS and pools separably -- 13 taps along x for the three nested windows at once, then 5 + 9 + 13
// taps along y over the row results -- instead of 169 global loads per output vector.  The row pass keeps the first
// maximum's dx, the column pass the first row reaching the maximum: together the first maximum in row-major window order,
// which is what ATen's max_pool2d backward routes the gradient to.
constexpr size_t SPP_PLANE_LDS = 64 * 1024;      // dynamic LDS available without opting in
template <typename T, bool IDX>
__global__ __launch_bounds__(256) void spp_fwd_plane_kernel(myolo_tensor x, myolo_tensor o5, myolo_tensor o9, myolo_tensor o13,
                                                            uint8_t* idx, int band_rows) {
  // blockIdx.y = a band of `band_rows` output rows (round 3: a batch-1 32x64 plane is 32 workgroups as a whole -- 4 bands x 32 channel
  // groups fill more of the chip); the tile holds the band plus up to 6 halo rows on each side
  constexpr int SEG = ET<T>::SEG;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int W = x.w, H = x.h, HWfull = H * W;
  const int yb0 = blockIdx.y * band_rows, yb1 = yb0 + band_rows < H ? yb0 + band_rows : H;     // output rows of this workgroup
  const int ty0 = yb0 - 6 > 0 ? yb0 - 6 : 0, ty1 = yb1 + 6 < H ? yb1 + 6 : H;                  // tile rows
  const int HW = (ty1 - ty0) * W;                                         // pixels in the tile
  const int cap = (band_rows + 12) * W;                                   // plane pitch (host allocates for this)
  T* tile = reinterpret_cast<T*>(smem);                                   // [cap][SEG]
  T* rv = tile + (size_t)cap * SEG;                                       // [3][cap][SEG] row maxima
  uint8_t* ri = reinterpret_cast<uint8_t*>(rv + (size_t)3 * cap * SEG);   // [3][cap][SEG] dx index of the row maximum (training only)
  const int G = x.c / SEG;
  const int n = blockIdx.x / G, cg = blockIdx.x - n * G;
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    const int y = ty0 + p / W, xx = p % W;
    *reinterpret_cast<uint4*>(tile + (size_t)p * SEG) = ldg16(vptr<T>(x, n, y, xx) + cg * SEG);
  }
  __syncthreads();
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    const int yl = p / W, xx = p - yl * W;
    float m5[SEG], m9[SEG], m13[SEG];
    int i5[SEG], i9[SEG], i13[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) { m5[i] = m9[i] = m13[i] = -INFINITY; i5[i] = i9[i] = i13[i] = 0; }
#pragma unroll
    for (int dx = -6; dx <= 6; ++dx) {
      const int ix = xx + dx;
      if (ix < 0 || ix >= W) continue;
      float f[SEG];
      Vec<T>::unpack(*reinterpret_cast<const uint4*>(tile + (size_t)(yl * W + ix) * SEG), f);
#pragma unroll
      for (int i = 0; i < SEG; ++i) {
        if (f[i] > m13[i]) { m13[i] = f[i]; i13[i] = dx + 6; }
        if (dx >= -4 && dx <= 4 && f[i] > m9[i]) { m9[i] = f[i]; i9[i] = dx + 4; }
        if (dx >= -2 && dx <= 2 && f[i] > m5[i]) { m5[i] = f[i]; i5[i] = dx + 2; }
      }
    }
    *reinterpret_cast<uint4*>(rv + ((size_t)0 * cap + p) * SEG) = Vec<T>::pack(m5);
    *reinterpret_cast<uint4*>(rv + ((size_t)1 * cap + p) * SEG) = Vec<T>::pack(m9);
    *reinterpret_cast<uint4*>(rv + ((size_t)2 * cap + p) * SEG) = Vec<T>::pack(m13);
    if (IDX) {                                       // (eval: no index planes -- their LDS is not even allocated)
#pragma unroll
      for (int i = 0; i < SEG; ++i) {
        ri[((size_t)0 * cap + p) * SEG + i] = (uint8_t)i5[i];
        ri[((size_t)1 * cap + p) * SEG + i] = (uint8_t)i9[i];
        ri[((size_t)2 * cap + p) * SEG + i] = (uint8_t)i13[i];
      }
    }
  }
  __syncthreads();
  const int64_t plane = (int64_t)x.n * HWfull * x.c;
  const int nout = (yb1 - yb0) * W;
  for (int p = threadIdx.x; p < nout; p += blockDim.x) {
    const int y = yb0 + p / W, xx = p % W;
    const int64_t e = ((int64_t)n * HWfull + (int64_t)y * W + xx) * x.c + cg * SEG;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      const int r = 2 + 2 * w, K = 2 * r + 1;
      float m[SEG];
      int id[SEG];
#pragma unroll
      for (int i = 0; i < SEG; ++i) { m[i] = -INFINITY; id[i] = 0; }
      for (int dy = -r; dy <= r; ++dy) {
        const int iy = y + dy;
        if (iy < 0 || iy >= H) continue;
        const size_t q = (size_t)w * cap + (iy - ty0) * W + xx;
        float f[SEG];
        Vec<T>::unpack(*reinterpret_cast<const uint4*>(rv + q * SEG), f);
#pragma unroll
        for (int i = 0; i < SEG; ++i)
          if (f[i] > m[i]) { m[i] = f[i]; if (IDX) id[i] = (dy + r) * K + ri[q * SEG + i]; }
      }
      const myolo_tensor& o = w == 0 ? o5 : (w == 1 ? o9 : o13);
      stg16(vptr<T>(o, n, y, xx) + cg * SEG, Vec<T>::pack(m));
      if (IDX) {
#pragma unroll
        for (int i = 0; i < SEG; ++i) idx[(int64_t)w * plane + e + i] = (uint8_t)id[i];
      }
    }
  }
}

// Eval, fp16 (the detect.py frame: no index planes): the same separable pooling on the PACKED halves -- max is exact in fp16, so the 13 + 27
// taps per output vector are 4 v_pk_max_f16 each instead of 8 unpacks + 8 compare / selects in fp32 (the plane kernel above was VALU-bound at
// batch 1: 22 us on the frame's critical path for a 1 MB map).  NaN: a NaN input loses against any number (as `f > m` above).
typedef _Float16 h2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 pkmax8(uint4 a, uint4 b) {
  uint4 r;
  unsigned int* ra = reinterpret_cast<unsigned int*>(&a);
  unsigned int* rb = reinterpret_cast<unsigned int*>(&b);
  unsigned int* rr = reinterpret_cast<unsigned int*>(&r);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h2v_t x = *reinterpret_cast<h2v_t*>(&ra[i]), y = *reinterpret_cast<h2v_t*>(&rb[i]);
    h2v_t m = __builtin_elementwise_max(x, y);
    rr[i] = *reinterpret_cast<unsigned int*>(&m);
  }
  return r;
}
__global__ __launch_bounds__(256) void spp_fwd_plane_h_kernel(myolo_tensor x, myolo_tensor o5, myolo_tensor o9, myolo_tensor o13, int band_rows) {
  constexpr int SEG = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int W = x.w, H = x.h;
  const int yb0 = blockIdx.y * band_rows, yb1 = yb0 + band_rows < H ? yb0 + band_rows : H;
  const int ty0 = yb0 - 6 > 0 ? yb0 - 6 : 0, ty1 = yb1 + 6 < H ? yb1 + 6 : H;
  const int HW = (ty1 - ty0) * W;
  const int cap = (band_rows + 12) * W;
  uint4* tile = reinterpret_cast<uint4*>(smem);                            // [cap]
  uint4* rv = tile + cap;                                                  // [3][cap] row maxima
  const int G = x.c / SEG;
  const int n = blockIdx.x / G, cg = blockIdx.x - n * G;
  const uint4 ninf = {0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u};
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    const int y = ty0 + p / W, xx = p % W;
    tile[p] = ldg16(vptr<half_t>(x, n, y, xx) + cg * SEG);
  }
  __syncthreads();
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    const int yl = p / W, xx = p - yl * W;
    uint4 m5 = ninf, m9 = ninf, m13 = ninf;
#pragma unroll
    for (int dx = -6; dx <= 6; ++dx) {
      const int ix = xx + dx;
      if (ix < 0 || ix >= W) continue;
      const uint4 v = tile[yl * W + ix];
      m13 = pkmax8(m13, v);
      if (dx >= -4 && dx <= 4) m9 = pkmax8(m9, v);
      if (dx >= -2 && dx <= 2) m5 = pkmax8(m5, v);
    }
    rv[p] = m5; rv[cap + p] = m9; rv[2 * cap + p] = m13;
  }
  __syncthreads();
  const int nout = (yb1 - yb0) * W;
  for (int p = threadIdx.x; p < nout; p += blockDim.x) {
    const int y = yb0 + p / W, xx = p % W;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      const int r = 2 + 2 * w;
      uint4 m = ninf;
      for (int dy = -r; dy <= r; ++dy) {
        const int iy = y + dy;
        if (iy < 0 || iy >= H) continue;
        m = pkmax8(m, rv[(size_t)w * cap + (iy - ty0) * W + xx]);
      }
      const myolo_tensor& o = w == 0 ? o5 : (w == 1 ? o9 : o13);
      stg16(vptr<half_t>(o, n, y, xx) + cg * SEG, m);
    }
  }
}

// transpose of the above: every output pixel adds its three gradients to the recorded arg-max positions of the plane (LDS fp32
// atomics), then the plane is written once.
template <typename T>
__global__ __launch_bounds__(256) void spp_bwd_plane_kernel(myolo_tensor g5, myolo_tensor g9, myolo_tensor g13,
                                                            const uint8_t* __restrict__ idx, myolo_tensor gx, int acc) {
  constexpr int SEG = ET<T>::SEG;
  extern __shared__ float accum[];                                        // [HW][SEG]
  const int HW = gx.h * gx.w, W = gx.w;
  const int G = gx.c / SEG;
  const int n = blockIdx.x / G, cg = blockIdx.x - n * G;
  for (int i = threadIdx.x; i < HW * SEG; i += blockDim.x) accum[i] = 0.f;
  __syncthreads();
  const int64_t plane = (int64_t)gx.n * HW * gx.c;
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    const int y = p / W, xx = p - y * W;
    const int64_t e = ((int64_t)n * HW + p) * gx.c + cg * SEG;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      const int r = 2 + 2 * w, K = 2 * r + 1;
      const myolo_tensor& g = w == 0 ? g5 : (w == 1 ? g9 : g13);
      float f[SEG];
      Vec<T>::unpack(ldg16(vptr<T>(g, n, y, xx) + cg * SEG), f);
#pragma unroll
      for (int i = 0; i < SEG; ++i) {
        const int id = idx[(int64_t)w * plane + e + i];
        const int ty = y + id / K - r, tx = xx + id % K - r;
        atomicAdd(accum + (size_t)(ty * W + tx) * SEG + i, f[i]);
      }
    }
  }
  __syncthreads();
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    const int y = p / W, xx = p - y * W;
    float a[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) a[i] = accum[(size_t)p * SEG + i];
    T* gp = vptr<T>(gx, n, y, xx) + cg * SEG;
    if (acc) {
      float o[SEG];
      Vec<T>::unpack(ldg16(gp), o);
#pragma unroll
      for (int i = 0; i < SEG; ++i) a[i] += o[i];
    }
    stg16(gp, Vec<T>::pack(a));
  }
}

// Round 6: the same transpose with the LANES ON THE CHANNELS.  The plane kernel above gives a thread one pixel x 8 channels: (1) its 16-byte
// loads are gx.c * 2 bytes apart (512 B at 256 channels: a quarter of every 64-byte sector is used -- the PMC counted 185 MB read for 22 MB of
// operands), and (2) neighbouring pixels usually share an arg-max position, so the lanes of a wave add to the SAME LDS address and the atomics
// serialise (90 us for a 4 MB tensor).  Here a workgroup owns CH = 32 consecutive channels of one image: 32 lanes read one pixel's 64 bytes
// (gradient) / 32 bytes (index) and add to 32 consecutive LDS words -- distinct banks whatever the arg-max pattern; 1024 threads = 32 runs of
// pixels per workgroup.
template <typename T, int CH, int NT>
__global__ __launch_bounds__(NT) void spp_bwd_chan_kernel(myolo_tensor g5, myolo_tensor g9, myolo_tensor g13,
                                                          const uint8_t* __restrict__ idx, myolo_tensor gx, int acc) {
  constexpr int SEG = ET<T>::SEG, NPL = NT / CH;
  extern __shared__ __attribute__((aligned(16))) float accum[];           // [HW][CH]
  const int HW = gx.h * gx.w, W = gx.w;
  const int GC = gx.c / CH;
  const int n = blockIdx.x / GC, c0 = (blockIdx.x - n * GC) * CH;
  for (int i = threadIdx.x; i < HW * CH / 4; i += NT) reinterpret_cast<float4*>(accum)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const int lc = threadIdx.x % CH, pl = threadIdx.x / CH;
  const int64_t plane = (int64_t)gx.n * HW * gx.c;
  const uint8_t* ib = idx + (int64_t)n * HW * gx.c + c0 + lc;
  const T* b5 = static_cast<const T*>(g5.ptr) + (int64_t)n * g5.sn + c0 + lc;
  const T* b9 = static_cast<const T*>(g9.ptr) + (int64_t)n * g9.sn + c0 + lc;
  const T* b13 = static_cast<const T*>(g13.ptr) + (int64_t)n * g13.sn + c0 + lc;
  // A thread walks RUN consecutive pixels of the row-major plane for its channel.  Neighbouring outputs mostly share their arg-max position (a
  // maximum owns the windows around it), so the gradient of a run of equal targets is summed in a register and reaches the LDS as ONE atomic:
  // fp32 LDS atomics are the kernel's bound (6.3 M of them took 44-67 us whatever the load schedule -- about one lane per 2-4 clocks and CU).
  // U pixels are loaded per step before the first use (clamped indices, zero weights past the end: no predicated loads).
  constexpr int U = 4;
  const int RUN = (HW + NPL - 1) / NPL;
  const int pbeg = pl * RUN;
  int cur[3] = {-1, -1, -1};
  float sm[3] = {0.f, 0.f, 0.f};
  for (int u0 = 0; u0 < RUN; u0 += U) {
    T f[U][3];
    uint8_t id[U][3];
    int yy[U], xc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int p = pbeg + u0 + u;
      p = p < HW ? p : HW - 1;
      const int y = p / W, xx = p - y * W;
      yy[u] = y; xc[u] = xx;
      f[u][0] = b5[(int64_t)y * g5.sh + (int64_t)xx * g5.sw];
      f[u][1] = b9[(int64_t)y * g9.sh + (int64_t)xx * g9.sw];
      f[u][2] = b13[(int64_t)y * g13.sh + (int64_t)xx * g13.sw];
      id[u][0] = ib[(int64_t)p * gx.c];
      id[u][1] = ib[plane + (int64_t)p * gx.c];
      id[u][2] = ib[2 * plane + (int64_t)p * gx.c];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool live = u0 + u < RUN && pbeg + u0 + u < HW;
#pragma unroll
      for (int w = 0; w < 3; ++w) {
        const int K = 5 + 4 * w, r = 2 + 2 * w;
        const int i = id[u][w];
        const int t = live ? (yy[u] + i / K - r) * W + xc[u] + i % K - r : -1;
        const float v = (float)f[u][w];
        if (t == cur[w]) {
          sm[w] += v;
        } else {
          if (cur[w] >= 0) atomicAdd(accum + (size_t)cur[w] * CH + lc, sm[w]);
          cur[w] = t; sm[w] = v;
        }
      }
    }
  }
#pragma unroll
  for (int w = 0; w < 3; ++w)
    if (cur[w] >= 0) atomicAdd(accum + (size_t)cur[w] * CH + lc, sm[w]);
  __syncthreads();
  constexpr int GP = CH / SEG;                                            // 16-byte pieces per pixel
  for (int v = threadIdx.x; v < HW * GP; v += NT) {
    const int p = v / GP, g = v - p * GP;
    const int y = p / W, xx = p - y * W;
    float a[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) a[i] = accum[(size_t)p * CH + g * SEG + i];
    T* gp = vptr<T>(gx, n, y, xx) + c0 + g * SEG;
    if (acc) {
      float o[SEG];
      Vec<T>::unpack(ldg16(gp), o);
#pragma unroll
      for (int i = 0; i < SEG; ++i) a[i] += o[i];
    }
    stg16(gp, Vec<T>::pack(a));
  }
}

// ---------------------------------------------------------------- copy / nearest upsample
template <typename T>
__global__ __launch_bounds__(256) void copy_up_fwd_kernel(myolo_tensor x, myolo_tensor out, int shift) {
  constexpr int SEG = ET<T>::SEG;
  const int G = out.c / SEG;
  const int64_t total = (int64_t)out.n * out.h * out.w * G;
  GRID_STRIDE(v, total) {
    int n, y, xx, cg;
    dec(v, G, out.w, out.h, n, y, xx, cg);
    stg16(vptr<T>(out, n, y, xx) + cg * SEG, ldg16(vptr<T>(x, n, y >> shift, xx >> shift) + cg * SEG));
  }
}
template <typename T>
__global__ __launch_bounds__(256) void copy_up_bwd_kernel(myolo_tensor gout, myolo_tensor gx, int shift, int acc) {
  constexpr int SEG = ET<T>::SEG;
  const int G = gx.c / SEG;
  const int64_t total = (int64_t)gx.n * gx.h * gx.w * G;
  const int r = 1 << shift;
  GRID_STRIDE(v, total) {
    int n, y, xx, cg;
    dec(v, G, gx.w, gx.h, n, y, xx, cg);
    float a[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) a[i] = 0.f;
    for (int dy = 0; dy < r; ++dy)
      for (int dx = 0; dx < r; ++dx) {
        float f[SEG];
        Vec<T>::unpack(ldg16(vptr<T>(gout, n, y * r + dy, xx * r + dx) + cg * SEG), f);
#pragma unroll
        for (int i = 0; i < SEG; ++i) a[i] += f[i];
      }
    T* gp = vptr<T>(gx, n, y, xx) + cg * SEG;
    if (acc) {
      float o[SEG];
      Vec<T>::unpack(ldg16(gp), o);
#pragma unroll
      for (int i = 0; i < SEG; ++i) a[i] += o[i];
    }
    stg16(gp, Vec<T>::pack(a));
  }
}

// ---------------------------------------------------------------- bilinear (align_corners=True)
// ATen upsample_bilinear2d: scale = (in-1)/(out-1) (0 if out==1); src = scale*dst; i0 = floor(src); i1 = min(i0+1,in-1);
// lambda1 = src - i0.
template <typename T>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(myolo_tensor x, myolo_tensor out, float sy, float sx) {
  constexpr int SEG = ET<T>::SEG;
  const int G = out.c / SEG;
  const int64_t total = (int64_t)out.n * out.h * out.w * G;
  GRID_STRIDE(v, total) {
    int n, y, xx, cg;
    dec(v, G, out.w, out.h, n, y, xx, cg);
    const float fy = sy * (float)y, fx = sx * (float)xx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + 1 < x.h ? y0 + 1 : x.h - 1, x1 = x0 + 1 < x.w ? x0 + 1 : x.w - 1;
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    float a[SEG], b[SEG], c[SEG], d[SEG], o[SEG];
    Vec<T>::unpack(ldg16(vptr<T>(x, n, y0, x0) + cg * SEG), a);
    Vec<T>::unpack(ldg16(vptr<T>(x, n, y0, x1) + cg * SEG), b);
    Vec<T>::unpack(ldg16(vptr<T>(x, n, y1, x0) + cg * SEG), c);
    Vec<T>::unpack(ldg16(vptr<T>(x, n, y1, x1) + cg * SEG), d);
#pragma unroll
    for (int i = 0; i < SEG; ++i)
      o[i] = (1.f - ly) * ((1.f - lx) * a[i] + lx * b[i]) + ly * ((1.f - lx) * c[i] + lx * d[i]);
    stg16(vptr<T>(out, n, y, xx) + cg * SEG, Vec<T>::pack(o));
  }
}

// gather form of the transpose: input pixel (iy,ix) collects every output whose 2x2 footprint touches it
__device__ __forceinline__ void out_range(int i, int in, int out, float s, int& lo, int& hi) {
  // outputs o with floor(s*o) in {i-1, i}  (i1 = i0+1 may equal i);  conservative bounds, exact test inside
  if (out == 1 || in == 1 || s <= 0.f) { lo = 0; hi = out - 1; return; }
  float a = ((float)i - 1.f) / s, b = ((float)i + 1.f) / s;
  lo = (int)floorf(a) - 1; hi = (int)ceilf(b) + 1;
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(myolo_tensor gout, myolo_tensor gx, float sy, float sx, int acc) {
  constexpr int SEG = ET<T>::SEG;
  const int G = gx.c / SEG;
  const int64_t total = (int64_t)gx.n * gx.h * gx.w * G;
  GRID_STRIDE(v, total) {
    int n, y, xx, cg;
    dec(v, G, gx.w, gx.h, n, y, xx, cg);
    int ylo, yhi, xlo, xhi;
    out_range(y, gx.h, gout.h, sy, ylo, yhi);
    out_range(xx, gx.w, gout.w, sx, xlo, xhi);
    float a[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) a[i] = 0.f;
    for (int oy = ylo; oy <= yhi; ++oy) {
      const float fy = sy * (float)oy;
      const int y0 = (int)fy;
      const int y1 = y0 + 1 < gx.h ? y0 + 1 : gx.h - 1;
      const float ly = fy - (float)y0;
      float wy = 0.f;
      if (y0 == y) wy += 1.f - ly;
      if (y1 == y) wy += ly;
      if (wy == 0.f) continue;                       // (a wave's lanes share y but for a row change: a uniform branch)
      // round 6: four output columns per step, their loads issued together (clamped column, weight 0 past the range) -- the per-column
      // `continue` kept every load next to its wait
      for (int ox0 = xlo; ox0 <= xhi; ox0 += 4) {
        uint4 raw[4];
        float wgt[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int oxr = ox0 + j, ox = oxr <= xhi ? oxr : xhi;
          raw[j] = ldg16(vptr<T>(gout, n, oy, ox) + cg * SEG);
          const float fx = sx * (float)ox;
          const int x0 = (int)fx;
          const int x1 = x0 + 1 < gx.w ? x0 + 1 : gx.w - 1;
          const float lx = fx - (float)x0;
          float wx = 0.f;
          if (x0 == xx) wx += 1.f - lx;
          if (x1 == xx) wx += lx;
          wgt[j] = oxr <= xhi ? wy * wx : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float f[SEG];
          Vec<T>::unpack(raw[j], f);
#pragma unroll
          for (int i = 0; i < SEG; ++i) a[i] += wgt[j] * f[i];
        }
      }
    }
    T* gp = vptr<T>(gx, n, y, xx) + cg * SEG;
    if (acc) {
      float o[SEG];
      Vec<T>::unpack(ldg16(gp), o);
#pragma unroll
      for (int i = 0; i < SEG; ++i) a[i] += o[i];
    }
    stg16(gp, Vec<T>::pack(a));
  }
}

// large-footprint variant (PyramidPooling: 1x1..6x6 -> 64x128, common.py:534-537): one workgroup per (input pixel,
// block of <=8 channel groups); the 256 threads stride over the pixel's output footprint and reduce through LDS.
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_big_kernel(myolo_tensor gout, myolo_tensor gx, float sy, float sx, int acc,
                                                               int gb) {
  constexpr int SEG = ET<T>::SEG;
  __shared__ float red[256 * 8];
  const int G = gx.c / SEG;
  const int nblk_c = (G + gb - 1) / gb;
  int b = blockIdx.x;
  const int cb = b % nblk_c; b /= nblk_c;
  const int xx = b % gx.w; b /= gx.w;
  const int y = b % gx.h; const int n = b / gx.h;
  const int cgl = threadIdx.x % gb, lane = threadIdx.x / gb, lanes = 256 / gb;
  const int cg = cb * gb + cgl;
  int ylo, yhi, xlo, xhi;
  out_range(y, gx.h, gout.h, sy, ylo, yhi);
  out_range(xx, gx.w, gout.w, sx, xlo, xhi);
  const int fw = xhi - xlo + 1, fp = (yhi - ylo + 1) * fw;
  float a[SEG];
#pragma unroll
  for (int i = 0; i < SEG; ++i) a[i] = 0.f;
  if (cg < G)
    for (int p = lane; p < fp; p += lanes) {
      const int oy = ylo + p / fw, ox = xlo + p % fw;
      const float fy = sy * (float)oy, fx = sx * (float)ox;
      const int y0 = (int)fy, x0 = (int)fx;
      const int y1 = y0 + 1 < gx.h ? y0 + 1 : gx.h - 1, x1 = x0 + 1 < gx.w ? x0 + 1 : gx.w - 1;
      const float ly = fy - (float)y0, lx = fx - (float)x0;
      float wy = 0.f, wx = 0.f;
      if (y0 == y) wy += 1.f - ly;
      if (y1 == y) wy += ly;
      if (x0 == xx) wx += 1.f - lx;
      if (x1 == xx) wx += lx;
      const float wgt = wy * wx;
      if (wgt == 0.f) continue;
      float f[SEG];
      Vec<T>::unpack(ldg16(vptr<T>(gout, n, oy, ox) + cg * SEG), f);
#pragma unroll
      for (int i = 0; i < SEG; ++i) a[i] += wgt * f[i];
    }
#pragma unroll
  for (int i = 0; i < SEG; ++i) red[threadIdx.x * 8 + i] = a[i];
  __syncthreads();
  for (int stride = lanes >> 1; stride > 0; stride >>= 1) {
    if (lane < stride)
#pragma unroll
      for (int i = 0; i < SEG; ++i) red[threadIdx.x * 8 + i] += red[(threadIdx.x + stride * gb) * 8 + i];
    __syncthreads();
  }
  if (lane == 0 && cg < G) {
    float o[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) o[i] = red[threadIdx.x * 8 + i];
    T* gp = vptr<T>(gx, n, y, xx) + cg * SEG;
    if (acc) {
      float q[SEG];
      Vec<T>::unpack(ldg16(gp), q);
#pragma unroll
      for (int i = 0; i < SEG; ++i) o[i] += q[i];
    }
    stg16(gp, Vec<T>::pack(o));
  }
}

// PyramidPooling footprints (1x1 .. 6x6 -> 64x128, thousands of outputs per input pixel), split variant: a workgroup owns one
// input row iy of one image and a chunk of <= BB_ROWS output rows of its footprint, staged in LDS with coalesced 16-byte loads.
// Thread (o = (ix, c), part) folds its share of the x footprint over all staged rows (the x weight is computed once per
// column); partial sums are combined in `scratch` (fp32 [n][kh][kw][C], zero on entry) with one atomic per thread and
// bilinear_bwd_finish converts.
constexpr int BB_MAXK = 6;
constexpr int BB_ROWS = 8;
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_split_kernel(myolo_tensor gout, int kh, int kw, float sy, float sx, int chunks,
                                                                 int per, float* scratch) {
  constexpr int SEG = ET<T>::SEG;
  extern __shared__ __attribute__((aligned(16))) char smem_bb[];
  T* rows = reinterpret_cast<T*>(smem_bb);          // [per][Wo][C]
  __shared__ float wys[BB_ROWS];
  const int C = gout.c, Wo = gout.w, G = C / SEG;
  int b = blockIdx.x;
  const int ch = b % chunks; b /= chunks;
  const int iy = b % kh; const int n = b / kh;
  int ylo, yhi;
  out_range(iy, kh, gout.h, sy, ylo, yhi);
  const int r0 = ylo + ch * per;
  int nr = yhi - r0 + 1;
  if (nr > per) nr = per;
  if (nr <= 0) return;
  if ((int)threadIdx.x < nr) {
    const int oy = r0 + threadIdx.x;
    const float fy = sy * (float)oy;
    const int y0 = (int)fy;
    const int y1 = y0 + 1 < kh ? y0 + 1 : kh - 1;
    const float ly = fy - (float)y0;
    float wy = 0.f;
    if (y0 == iy) wy += 1.f - ly;
    if (y1 == iy) wy += ly;
    wys[threadIdx.x] = wy;
  }
  const int vec_per_row = Wo * G;
  for (int v = threadIdx.x; v < nr * vec_per_row; v += 256) {
    const int r = v / vec_per_row, q = v - r * vec_per_row;
    const int px = q / G, cg = q - px * G;
    *reinterpret_cast<uint4*>(rows + ((size_t)r * Wo + px) * C + cg * SEG) = ldg16(vptr<T>(gout, n, r0 + r, px) + cg * SEG);
  }
  __syncthreads();
  const int nout = kw * C;
  const int parts = 256 / nout > 0 ? 256 / nout : 1;
  for (int o0 = 0; o0 < nout; o0 += 256) {               // nout > 256: several passes with parts = 1
    const int o = o0 + (int)threadIdx.x % (parts > 1 ? nout : 256);
    const int part = parts > 1 ? (int)threadIdx.x / nout : 0;
    if (o >= nout || part >= parts) continue;
    const int ix = o / C, c = o - ix * C;
    int xlo, xhi;
    out_range(ix, kw, Wo, sx, xlo, xhi);
    const int span = (xhi - xlo + 1 + parts - 1) / parts;
    const int xa = xlo + part * span;
    int xb = xa + span - 1;
    if (xb > xhi) xb = xhi;
    float acc = 0.f;
    for (int ox = xa; ox <= xb; ++ox) {
      const float fx = sx * (float)ox;
      const int x0 = (int)fx;
      const int x1 = x0 + 1 < kw ? x0 + 1 : kw - 1;
      const float lx = fx - (float)x0;
      float wx = 0.f;
      if (x0 == ix) wx += 1.f - lx;
      if (x1 == ix) wx += lx;
      if (wx == 0.f) continue;
      float sacc = 0.f;
      for (int r = 0; r < nr; ++r) sacc = fmaf(wys[r], (float)rows[((size_t)r * Wo + ox) * C + c], sacc);
      acc = fmaf(wx, sacc, acc);
    }
    if (acc != 0.f) atomicAdd(scratch + ((size_t)(n * kh + iy) * kw) * C + o, acc);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_finish_kernel(myolo_tensor gx, const float* __restrict__ scratch, int acc) {
  constexpr int SEG = ET<T>::SEG;
  const int G = gx.c / SEG;
  const int64_t total = (int64_t)gx.n * gx.h * gx.w * G;
  GRID_STRIDE(v, total) {
    int n, y, xx, cg;
    dec(v, G, gx.w, gx.h, n, y, xx, cg);
    const float* sp = scratch + (((size_t)n * gx.h + y) * gx.w + xx) * gx.c + cg * SEG;
    float a[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) a[i] = sp[i];
    T* gp = vptr<T>(gx, n, y, xx) + cg * SEG;
    if (acc) {
      float o[SEG];
      Vec<T>::unpack(ldg16(gp), o);
#pragma unroll
      for (int i = 0; i < SEG; ++i) a[i] += o[i];
    }
    stg16(gp, Vec<T>::pack(a));
  }
}

// ---------------------------------------------------------------- PyramidPooling: the four upsample backwards in one pass
// common.py:534-537 upsamples four k x k maps (k = 1, 2, 3, 6) to the feature map and concatenates them: in the backward their
// output gradients are four ADJACENT channel slices of one buffer.  One pass over those channels instead of four kernels that each
// reduce an 8 MB map to k x k values (0.2 TB/s: the outputs are too few to spread the work): a workgroup owns a strip of rows of one
// image, thread = (channel group -> branch, lane of 16 pixels); per row the x-fold into the <= 6 column cells runs in registers, rows
// are folded into the two row cells around them ((1-ly), ly), and when the walk leaves a row cell (or the strip ends) the 16 lanes of a
// channel group butterfly-reduce and add to the fp32 scratch (one atomic per (cell, channel) and workgroup).
struct PyrUp { myolo_tensor gout; int k[4]; int cgs_per_branch, nbranch, rows; float sy[4], sx[4]; int off[4]; };

template <typename T>
__global__ __launch_bounds__(256) void pyr_up_bwd_kernel(PyrUp p, float* __restrict__ scratch) {
  constexpr int SEG = ET<T>::SEG;
  constexpr int KMAX = 6;
  const int lane16 = threadIdx.x & 15, cg = threadIdx.x >> 4;
  const int t = cg / p.cgs_per_branch;
  const int strips = (p.gout.h + p.rows - 1) / p.rows;
  const int n = blockIdx.x / strips, ys = (blockIdx.x - n * strips) * p.rows;
  if (t >= p.nbranch) return;
  const int k = p.k[t];
  const float sy = p.sy[t], sx = p.sx[t];
  const int cb = (cg - t * p.cgs_per_branch) * SEG;              // first channel of this thread inside its branch
  const int cbr = p.cgs_per_branch * SEG;                        // channels per branch
  float top[KMAX][SEG], bot[KMAX][SEG];
#pragma unroll
  for (int j = 0; j < KMAX; ++j)
#pragma unroll
    for (int i = 0; i < SEG; ++i) { top[j][i] = 0.f; bot[j][i] = 0.f; }
  // scratch layout: [branch offset][n][k][k][channels of the branch]
  auto flush = [&](int row, float (&a)[KMAX][SEG]) {
    float* base = scratch + p.off[t] + (((size_t)n * k + row) * k) * cbr + cb;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (j >= k) break;
#pragma unroll
      for (int i = 0; i < SEG; ++i) {
        float v = a[j][i];
        v += __shfl_xor(v, 1, 16); v += __shfl_xor(v, 2, 16); v += __shfl_xor(v, 4, 16); v += __shfl_xor(v, 8, 16);
        if (((j * SEG + i) & 15) == lane16 && v != 0.f) atomicAdd(base + (size_t)j * cbr + i, v);
        a[j][i] = 0.f;
      }
    }
  };
  const int yend = ys + p.rows < p.gout.h ? ys + p.rows : p.gout.h;
  int icur = (int)(sy * (float)ys);
  const T* gb = reinterpret_cast<const T*>(p.gout.ptr) + (int64_t)n * p.gout.sn + (int64_t)cg * SEG;
  for (int y = ys; y < yend; ++y) {
    const float fy = sy * (float)y;
    const int i0 = (int)fy;
    const float ly = fy - (float)i0;
    while (icur != i0) {                       // (uniform over the 16 lanes of a channel group)
      flush(icur, top);
#pragma unroll
      for (int j = 0; j < KMAX; ++j)
#pragma unroll
        for (int i = 0; i < SEG; ++i) { top[j][i] = bot[j][i]; bot[j][i] = 0.f; }
      ++icur;
    }
    float row[KMAX][SEG];
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
#pragma unroll
      for (int i = 0; i < SEG; ++i) row[j][i] = 0.f;
    const T* rp = gb + (int64_t)y * p.gout.sh;
    for (int xq = lane16; xq < p.gout.w; xq += 64) {         // four pixels' loads in flight (round 6; clamped column, weight 0 past the row)
      uint4 raw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int x = xq + 16 * u;
        raw[u] = ldg16(rp + (int64_t)(x < p.gout.w ? x : p.gout.w - 1) * p.gout.sw);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int x = xq + 16 * u;
        float f[SEG];
        Vec<T>::unpack(raw[u], f);
        const float live = x < p.gout.w ? 1.f : 0.f;
        const float fx = sx * (float)x;
        const int j0 = (int)fx;
        const float lx = fx - (float)j0;
        const int j1 = j0 + 1 < k ? j0 + 1 : k - 1;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
          const float w = ((j == j0 ? 1.f - lx : 0.f) + (j == j1 ? lx : 0.f)) * live;
#pragma unroll
          for (int i = 0; i < SEG; ++i) row[j][i] += w * f[i];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
#pragma unroll
      for (int i = 0; i < SEG; ++i) { top[j][i] += (1.f - ly) * row[j][i]; bot[j][i] += ly * row[j][i]; }
  }
  flush(icur, top);
  flush(icur + 1 < k ? icur + 1 : k - 1, bot);
}

// forward of the same four upsamples: one launch writes the four adjacent channel slices (bilinear_fwd_kernel's arithmetic per branch)
struct PyrUpF { myolo_tensor x[4]; myolo_tensor out; int cgs_per_branch, n; float sy[4], sx[4]; };
template <typename T>
__global__ __launch_bounds__(256) void pyr_up_fwd_kernel(PyrUpF p) {
  constexpr int SEG = ET<T>::SEG;
  const int G = p.out.c / SEG;
  const int64_t total = (int64_t)p.out.n * p.out.h * p.out.w * G;
  GRID_STRIDE(v, total) {
    int n, y, xx, cg;
    dec(v, G, p.out.w, p.out.h, n, y, xx, cg);
    const int t = cg / p.cgs_per_branch, cl = (cg - t * p.cgs_per_branch) * SEG;
    const myolo_tensor& x = p.x[t];
    const float fy = p.sy[t] * (float)y, fx = p.sx[t] * (float)xx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + 1 < x.h ? y0 + 1 : x.h - 1, x1 = x0 + 1 < x.w ? x0 + 1 : x.w - 1;
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    float a[SEG], b[SEG], c[SEG], d[SEG], o[SEG];
    Vec<T>::unpack(ldg16(vptr<T>(x, n, y0, x0) + cl), a);
    Vec<T>::unpack(ldg16(vptr<T>(x, n, y0, x1) + cl), b);
    Vec<T>::unpack(ldg16(vptr<T>(x, n, y1, x0) + cl), c);
    Vec<T>::unpack(ldg16(vptr<T>(x, n, y1, x1) + cl), d);
#pragma unroll
    for (int i = 0; i < SEG; ++i)
      o[i] = (1.f - ly) * ((1.f - lx) * a[i] + lx * b[i]) + ly * ((1.f - lx) * c[i] + lx * d[i]);
    stg16(vptr<T>(p.out, n, y, xx) + cg * SEG, Vec<T>::pack(o));
  }
}

struct PyrFin { myolo_tensor gx[4]; int acc[4]; int off[4]; int n; };
template <typename T>
__global__ __launch_bounds__(256) void pyr_up_bwd_finish_kernel(PyrFin p, const float* __restrict__ scratch) {
  constexpr int SEG = ET<T>::SEG;
  for (int t = 0; t < p.n; ++t) {
    const myolo_tensor& gx = p.gx[t];
    const int G = gx.c / SEG;
    const int64_t total = (int64_t)gx.n * gx.h * gx.w * G;
    GRID_STRIDE(v, total) {
      int n, y, xx, cg;
      dec(v, G, gx.w, gx.h, n, y, xx, cg);
      const float* sp = scratch + p.off[t] + (((size_t)n * gx.h + y) * gx.w + xx) * gx.c + cg * SEG;
      float a[SEG];
#pragma unroll
      for (int i = 0; i < SEG; ++i) a[i] = sp[i];
      T* gp = vptr<T>(gx, n, y, xx) + cg * SEG;
      if (p.acc[t]) {
        float o[SEG];
        Vec<T>::unpack(ldg16(gp), o);
#pragma unroll
        for (int i = 0; i < SEG; ++i) a[i] += o[i];
      }
      stg16(gp, Vec<T>::pack(a));
    }
  }
}

// ---------------------------------------------------------------- adaptive average pool
// bin i of k over H: [floor(i*H/k), ceil((i+1)*H/k))
template <typename T>
__global__ __launch_bounds__(256) void aap_fwd_kernel(myolo_tensor x, myolo_tensor out) {
  constexpr int SEG = ET<T>::SEG;
  // one workgroup per (n, by, bx); threads = channel groups x pixel lanes
  __shared__ float red[256 * 8];
  const int G = x.c / SEG;
  const int kb = out.h, kw = out.w;
  int b = blockIdx.x;
  const int bx = b % kw; b /= kw;
  const int by = b % kb; const int n = b / kb;
  const int y0 = (by * x.h) / kb, y1 = ((by + 1) * x.h + kb - 1) / kb;
  const int x0 = (bx * x.w) / kw, x1 = ((bx + 1) * x.w + kw - 1) / kw;
  const int bw = x1 - x0, npix = (y1 - y0) * bw;
  for (int cg0 = 0; cg0 < G; cg0 += 256) {
    const int gcount = (G - cg0) < 256 ? (G - cg0) : 256;
    const int lanes = 256 / gcount;                    // pixel lanes per channel group
    const int cg = cg0 + threadIdx.x % gcount, pl = threadIdx.x / gcount;
    float a[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) a[i] = 0.f;
    if (pl < lanes)
      for (int p = pl; p < npix; p += lanes) {
        const int yy = y0 + p / bw, xx = x0 + p % bw;
        float f[SEG];
        Vec<T>::unpack(ldg16(vptr<T>(x, n, yy, xx) + cg * SEG), f);
#pragma unroll
        for (int i = 0; i < SEG; ++i) a[i] += f[i];
      }
#pragma unroll
    for (int i = 0; i < SEG; ++i) red[threadIdx.x * 8 + i] = a[i];
    __syncthreads();
    if (threadIdx.x < gcount) {
      float s[SEG];
#pragma unroll
      for (int i = 0; i < SEG; ++i) s[i] = 0.f;
      for (int q = 0; q < lanes; ++q)
#pragma unroll
        for (int i = 0; i < SEG; ++i) s[i] += red[(q * gcount + threadIdx.x) * 8 + i];
      const float inv = 1.f / (float)npix;
#pragma unroll
      for (int i = 0; i < SEG; ++i) s[i] *= inv;
      stg16(vptr<T>(out, n, by, bx) + (cg0 + threadIdx.x) * SEG, Vec<T>::pack(s));
    }
    __syncthreads();
  }
}
// split variant for big bins (PyramidPooling k=1..3 on a 64x128 map: thousands of pixels per bin but only n*k*k bins): every
// bin is cut into gridDim.y pixel chunks; partial sums go to an fp32 scratch (zeroed by the caller) with one atomic per
// channel per workgroup, aap_finish_kernel scales and casts.
template <typename T>
__global__ __launch_bounds__(256) void aap_fwd_split_kernel(myolo_tensor x, int kb, int kw, float* scratch) {
  constexpr int SEG = ET<T>::SEG;
  __shared__ float red[256 * 8];
  const int G = x.c / SEG;
  int b = blockIdx.x;
  const int bx = b % kw; b /= kw;
  const int by = b % kb; const int n = b / kb;
  const int y0 = (by * x.h) / kb, y1 = ((by + 1) * x.h + kb - 1) / kb;
  const int x0 = (bx * x.w) / kw, x1 = ((bx + 1) * x.w + kw - 1) / kw;
  const int bw = x1 - x0, npix = (y1 - y0) * bw;
  const int chunk = (npix + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * chunk;
  const int p1 = p0 + chunk < npix ? p0 + chunk : npix;
  float* dst = scratch + ((int64_t)(n * kb + by) * kw + bx) * x.c;
  for (int cg0 = 0; cg0 < G; cg0 += 256) {
    const int gcount = (G - cg0) < 256 ? (G - cg0) : 256;
    const int lanes = 256 / gcount;
    const int cg = cg0 + threadIdx.x % gcount, pl = threadIdx.x / gcount;
    float a[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) a[i] = 0.f;
    if (pl < lanes)
      for (int p = p0 + pl; p < p1; p += lanes) {
        const int yy = y0 + p / bw, xx = x0 + p % bw;
        float f[SEG];
        Vec<T>::unpack(ldg16(vptr<T>(x, n, yy, xx) + cg * SEG), f);
#pragma unroll
        for (int i = 0; i < SEG; ++i) a[i] += f[i];
      }
#pragma unroll
    for (int i = 0; i < SEG; ++i) red[threadIdx.x * 8 + i] = a[i];
    __syncthreads();
    if (threadIdx.x < gcount) {
#pragma unroll
      for (int i = 0; i < SEG; ++i) {
        float sm = 0.f;
        for (int q = 0; q < lanes; ++q) sm += red[(q * gcount + threadIdx.x) * 8 + i];
        atomicAdd(dst + (cg0 + threadIdx.x) * SEG + i, sm);
      }
    }
    __syncthreads();
  }
}
template <typename T>
__global__ __launch_bounds__(256) void aap_finish_kernel(myolo_tensor x, myolo_tensor out, const float* scratch) {
  const int kb = out.h, kw = out.w;
  const int64_t total = (int64_t)out.n * kb * kw * out.c;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % out.c); int64_t r = i / out.c;
    const int bx = (int)(r % kw); r /= kw;
    const int by = (int)(r % kb); const int n = (int)(r / kb);
    const int y0 = (by * x.h) / kb, y1 = ((by + 1) * x.h + kb - 1) / kb;
    const int x0 = (bx * x.w) / kw, x1 = ((bx + 1) * x.w + kw - 1) / kw;
    vptr<T>(out, n, by, bx)[c] = (T)(scratch[i] / (float)((y1 - y0) * (x1 - x0)));
  }
}


// ---- PyramidPooling's pools (AdaptiveAvgPool2d(1), (2), (3), (6) of the SAME map, common.py:521-524) in ONE pass over x ----------------
// Four separate passes read the 128-channel map four times (4 x 29 us for the 128x256 map of a 2048x1024 frame, plus four finish launches).
// Here a workgroup owns a strip of rows; a thread owns 8 channels (one 16-byte vector) and a contiguous x range of W/PL pixels, which is
// narrower than any bin, so per pool it touches at most three horizontal bins: three register accumulators per pool, added to the
// workgroup's [bins][C] fp32 table in LDS once per row (rows map to one or two vertical bins), the table goes to the global scratch with
// one atomic per touched (bin, channel).  Bin b of k over H is [floor(b*H/k), ceil((b+1)*H/k)): pixel y lies in bin y*k/H and, on a
// fractional boundary, in ONE neighbour.
constexpr int AAP_MAXP = 4;
struct AapFwdMulti { int np; int k[AAP_MAXP]; int bin0[AAP_MAXP]; int nbins; };

// bins of `k` over `n` that contain coordinate v: b (always) and nb (-1 if none)
__device__ __forceinline__ void aap_bins(int v, int n, int k, int& b, int& nb) {
  b = (int)(((uint32_t)v * (uint32_t)k) / (uint32_t)n);
  nb = -1;
  if (b > 0 && v < ((b * n + k - 1) / k)) nb = b - 1;                 // v < ceil(b*n/k): still inside bin b-1
  else if (b + 1 < k && v >= ((b + 1) * n) / k) nb = b + 1;           // v >= floor((b+1)*n/k): already inside bin b+1
}

constexpr int AAP_REPL = 8;     // replicas of the global bin table: workgroup w adds into replica w % 8 (same-address fp32 atomics retire
                                // at ~110 ns each: 256 workgroups on the 128 addresses of a global-average bin were 28 us of nothing else)
template <typename T>
__global__ __launch_bounds__(256) void aap_fwd_multi_kernel(myolo_tensor x, AapFwdMulti mp, int rows_per_wg, float* scratch) {
  constexpr int SEG = ET<T>::SEG;
  extern __shared__ float sbin[];                                      // [nbins][C]
  const int C = x.c, G = C / SEG, PL = 256 / G;
  const int cg = threadIdx.x % G, pl = threadIdx.x / G;
  const int n = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_wg, r1 = r0 + rows_per_wg < x.h ? r0 + rows_per_wg : x.h;
  for (int i = threadIdx.x; i < mp.nbins * C; i += 256) sbin[i] = 0.f;
  __syncthreads();
  const int xa = (int)(((int64_t)pl * x.w) / PL), xb = (int)(((int64_t)(pl + 1) * x.w) / PL);
  // the (at most three) horizontal bins per pool this thread's x range can touch, as pixel intervals [lo, hi): no division per pixel
  int bA[AAP_MAXP], lo[AAP_MAXP][3], hi[AAP_MAXP][3];
#pragma unroll
  for (int p = 0; p < AAP_MAXP; ++p) {
    int b = 0, nb = -1;
    const int k = p < mp.np ? mp.k[p] : 1;
    if (p < mp.np && xa < xb) aap_bins(xa, x.w, k, b, nb);
    bA[p] = (nb >= 0 && nb < b) ? nb : b;                             // lowest bin the range can touch
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int bb = bA[p] + j;
      const bool ok = p < mp.np && bb < k;
      lo[p][j] = ok ? (bb * x.w) / k : 0x7fffffff;
      hi[p][j] = ok ? ((bb + 1) * x.w + k - 1) / k : 0;
    }
  }
  // Round 6: the row sums stay in registers while a pool's VERTICAL bin (pair) does not change and reach the LDS table once per run of rows, not
  // once per row: fp32 LDS atomics retire at about one lane per 2-4 clocks and CU (spp_bwd_chan_kernel above), and ~50 of them per thread and
  // row were this kernel's time (the loads were not: round 5 measured four in flight neutral).  A bin of the 64-row map spans 11-64 rows, a
  // workgroup's strip 4: one flush per pool and workgroup instead of four, unless a bin boundary crosses the strip.
  float acc[AAP_MAXP][3][SEG];
  int cby[AAP_MAXP], cnby[AAP_MAXP];
#pragma unroll
  for (int p = 0; p < AAP_MAXP; ++p) {
    cby[p] = -1; cnby[p] = -1;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int i = 0; i < SEG; ++i) acc[p][j][i] = 0.f;
  }
  auto flush = [&](int p) {          // (p is a constant after unrolling: acc stays in registers)
    if (cby[p] < 0) return;
    const int k = mp.k[p];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int bx = bA[p] + j;
      if (!(bx >= k || hi[p][j] <= xa || lo[p][j] >= xb)) {           // bin touched by this thread's range
        float* d0 = sbin + (mp.bin0[p] + cby[p] * k + bx) * C + cg * SEG;
#pragma unroll
        for (int i = 0; i < SEG; ++i) atomicAdd(d0 + i, acc[p][j][i]);
        if (cnby[p] >= 0) {
          float* d1 = sbin + (mp.bin0[p] + cnby[p] * k + bx) * C + cg * SEG;
#pragma unroll
          for (int i = 0; i < SEG; ++i) atomicAdd(d1 + i, acc[p][j][i]);
        }
      }
#pragma unroll
      for (int i = 0; i < SEG; ++i) acc[p][j][i] = 0.f;
    }
  };
  for (int y = r0; y < r1; ++y) {
#pragma unroll
    for (int p = 0; p < AAP_MAXP; ++p) {
      if (p < mp.np) {
        int by, nby;
        aap_bins(y, x.h, mp.k[p], by, nby);
        if (by != cby[p] || nby != cnby[p]) { flush(p); cby[p] = by; cnby[p] = nby; }
      }
    }
    // three pixels' loads in flight (clamped column; a pixel past the range matches no bin interval).  Round 5 measured this neutral -- while the
    // per-row LDS atomics were the bound; with those gone the dependent loads are what is left (36.6 us per launch for a 33 MB read)
    constexpr int U = 3;            // (four: 242 VGPRs and a spill)
    for (int x0 = xa; x0 < xb; x0 += U) {
      uint4 raw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) raw[u] = ldg16(vptr<T>(x, n, y, x0 + u < xb ? x0 + u : xb - 1) + cg * SEG);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int xx = x0 + u < xb ? x0 + u : 0x7ffffff0;
        float f[SEG];
        Vec<T>::unpack(raw[u], f);
#pragma unroll
        for (int p = 0; p < AAP_MAXP; ++p)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const float m = (xx >= lo[p][j] && xx < hi[p][j]) ? 1.f : 0.f;
#pragma unroll
            for (int i = 0; i < SEG; ++i) acc[p][j][i] += m * f[i];
          }
      }
    }
  }
#pragma unroll
  for (int p = 0; p < AAP_MAXP; ++p) flush(p);
  __syncthreads();
  const int rep = (blockIdx.x + blockIdx.y) % AAP_REPL;
  float* dst = scratch + ((int64_t)rep * gridDim.y + n) * mp.nbins * C;
  for (int i = threadIdx.x; i < mp.nbins * C; i += 256) {
    const float v = sbin[i];
    if (v != 0.f) atomicAdd(dst + i, v);
  }
}
// scale + cast of every pool's bins: outs[p] [n, k, k, C]
struct AapOuts { myolo_tensor o[AAP_MAXP]; };
template <typename T>
__global__ __launch_bounds__(256) void aap_finish_multi_kernel(int H, int W, int C, int N, AapFwdMulti mp, AapOuts outs, const float* scratch) {
  const int64_t total = (int64_t)N * mp.nbins * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C); int64_t r = i / C;
    const int bin = (int)(r % mp.nbins); const int n = (int)(r / mp.nbins);
    int p = 0;
#pragma unroll
    for (int q = 1; q < AAP_MAXP; ++q) if (q < mp.np && bin >= mp.bin0[q]) p = q;
    const int k = mp.k[p], lb = bin - mp.bin0[p];
    const int by = lb / k, bx = lb - by * k;
    const int y0 = (by * H) / k, y1 = ((by + 1) * H + k - 1) / k;
    const int x0 = (bx * W) / k, x1 = ((bx + 1) * W + k - 1) / k;
    float sum = 0.f;
#pragma unroll
    for (int rp = 0; rp < AAP_REPL; ++rp) sum += scratch[(int64_t)rp * total + i];
    vptr<T>(outs.o[p], n, by, bx)[c] = (T)(sum / (float)((y1 - y0) * (x1 - x0)));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void aap_bwd_kernel(myolo_tensor gout, myolo_tensor gx, int acc) {
  constexpr int SEG = ET<T>::SEG;
  const int G = gx.c / SEG;
  const int kb = gout.h, kw = gout.w;
  const int64_t total = (int64_t)gx.n * gx.h * gx.w * G;
  GRID_STRIDE(v, total) {
    int n, y, xx, cg;
    dec(v, G, gx.w, gx.h, n, y, xx, cg);
    float a[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) a[i] = 0.f;
    // bins [floor(b*H/k), ceil((b+1)*H/k)) overlap by at most one pixel: only the bin y*k/H lands in and its two neighbours can hold y
    const int byc = (int)(((uint32_t)y * (uint32_t)kb) / (uint32_t)gx.h), bxc = (int)(((uint32_t)xx * (uint32_t)kw) / (uint32_t)gx.w);
    for (int by = byc > 0 ? byc - 1 : 0; by <= byc + 1 && by < kb; ++by) {
      const int y0 = (by * gx.h) / kb, y1 = ((by + 1) * gx.h + kb - 1) / kb;
      if (y < y0 || y >= y1) continue;
      for (int bx = bxc > 0 ? bxc - 1 : 0; bx <= bxc + 1 && bx < kw; ++bx) {
        const int x0 = (bx * gx.w) / kw, x1 = ((bx + 1) * gx.w + kw - 1) / kw;
        if (xx < x0 || xx >= x1) continue;
        float f[SEG];
        Vec<T>::unpack(ldg16(vptr<T>(gout, n, by, bx) + cg * SEG), f);
        const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
#pragma unroll
        for (int i = 0; i < SEG; ++i) a[i] += f[i] * inv;
      }
    }
    T* gp = vptr<T>(gx, n, y, xx) + cg * SEG;
    if (acc) {
      float o[SEG];
      Vec<T>::unpack(ldg16(gp), o);
#pragma unroll
      for (int i = 0; i < SEG; ++i) a[i] += o[i];
    }
    stg16(gp, Vec<T>::pack(a));
  }
}

// the same for up to four pooled maps of ONE input (PyramidPooling's bins 1, 2, 3, 6 -- common.py:521-524): the input gradient is
// read-modified-written once instead of once per bin (4 x 66 MB at 16x64x128x128)
struct AapMulti { myolo_tensor g[4]; int n; };
// the bins [floor(b*H/k), ceil((b+1)*H/k)) a row / column belongs to (one or two per pool) are tabulated in LDS once per workgroup:
// the per-pixel integer divisions of aap_bwd_kernel (four pools x up to nine candidate bins) were 2/3 of this kernel's time
template <typename T>
__global__ __launch_bounds__(256) void aap_bwd_multi_kernel(AapMulti m, myolo_tensor gx, int acc) {
  constexpr int SEG = ET<T>::SEG;
  extern __shared__ unsigned char tabs[];                 // [4][H] row: first bin | count << 4 ; [4][W] column ; then extents
  unsigned char* rowt = tabs;
  unsigned char* colt = tabs + 4 * gx.h;
  unsigned short* exth = reinterpret_cast<unsigned short*>(colt + 4 * gx.w);   // [4][8] bin heights, [4][8] bin widths (k <= 8)
  unsigned short* extw = exth + 32;
  for (int i = threadIdx.x; i < 4 * (gx.h + gx.w); i += blockDim.x) {
    const bool isrow = i < 4 * gx.h;
    const int j = isrow ? i : i - 4 * gx.h;
    const int L = isrow ? gx.h : gx.w;
    const int t = j / L, p = j - t * L;
    unsigned char v = 0;
    if (t < m.n) {
      const int k = isrow ? m.g[t].h : m.g[t].w;
      const int bc = (int)(((uint32_t)p * (uint32_t)k) / (uint32_t)L);
      int first = -1, cnt = 0;
      for (int bb = bc > 0 ? bc - 1 : 0; bb <= bc + 1 && bb < k; ++bb) {
        const int p0 = (bb * L) / k, p1 = ((bb + 1) * L + k - 1) / k;
        if (p >= p0 && p < p1) { if (first < 0) first = bb; ++cnt; }
      }
      v = (unsigned char)(first | (cnt << 4));
    }
    (isrow ? rowt : colt)[j] = v;
  }
  for (int i = threadIdx.x; i < 64; i += blockDim.x) {
    const bool ish = i < 32;
    const int j = ish ? i : i - 32, t = j >> 3, bb = j & 7;
    unsigned short v = 1;
    if (t < m.n) {
      const int k = ish ? m.g[t].h : m.g[t].w, L = ish ? gx.h : gx.w;
      if (bb < k) v = (unsigned short)(((bb + 1) * L + k - 1) / k - (bb * L) / k);
    }
    (ish ? exth : extw)[j] = v;
  }
  __syncthreads();
  const int G = gx.c / SEG;
  const int64_t total = (int64_t)gx.n * gx.h * gx.w * G;
  GRID_STRIDE(v, total) {
    int n, y, xx, cg;
    dec(v, G, gx.w, gx.h, n, y, xx, cg);
    // round 6: the loads every pixel needs -- its first bin of each pool and the old gradient -- are issued together before the first use; the
    // second bin of an axis (only pixels on a fractional bin boundary have one) keeps the loop.  (All 2 x 2 candidates unconditionally with zero
    // weights measured SLOWER, 81 against 63 us: sixteen 16-byte vectors to unpack and scale per pixel made the pass VALU-bound.)
    T* gp = vptr<T>(gx, n, y, xx) + cg * SEG;
    uint4 old = {0u, 0u, 0u, 0u};
    if (acc) old = ldg16(gp);
    uint4 raw[4];
    float w0[4];
    int by0[4], nby[4], bx0[4], nbx[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bool on = t < m.n;
      const int tt = on ? t : 0;
      const int rv = rowt[tt * gx.h + y], cv = colt[tt * gx.w + xx];
      by0[t] = rv & 15; nby[t] = on ? rv >> 4 : 0; bx0[t] = cv & 15; nbx[t] = on ? cv >> 4 : 0;
      raw[t] = ldg16(vptr<T>(m.g[tt], n, by0[t], bx0[t]) + cg * SEG);
      w0[t] = on ? 1.f / (float)((int)exth[tt * 8 + by0[t]] * (int)extw[tt * 8 + bx0[t]]) : 0.f;
    }
    float a[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) a[i] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float f[SEG];
      Vec<T>::unpack(raw[t], f);
#pragma unroll
      for (int i = 0; i < SEG; ++i) a[i] += f[i] * w0[t];
      if (nby[t] * nbx[t] > 1) {                                     // a boundary pixel: the other one to three bins
        for (int j = 0; j < nby[t]; ++j)
          for (int q = 0; q < nbx[t]; ++q) {
            if (j == 0 && q == 0) continue;
            const int by = by0[t] + j, bx = bx0[t] + q;
            float f2[SEG];
            Vec<T>::unpack(ldg16(vptr<T>(m.g[t], n, by, bx) + cg * SEG), f2);
            const float inv = 1.f / (float)((int)exth[t * 8 + by] * (int)extw[t * 8 + bx]);
#pragma unroll
            for (int i = 0; i < SEG; ++i) a[i] += f2[i] * inv;
          }
      }
    }
    if (acc) {
      float o[SEG];
      Vec<T>::unpack(old, o);
#pragma unroll
      for (int i = 0; i < SEG; ++i) a[i] += o[i];
    }
    stg16(gp, Vec<T>::pack(a));
  }
}

// ---------------------------------------------------------------- FFM gate
template <typename T>
__global__ __launch_bounds__(256) void gate_fwd_kernel(myolo_tensor feat, myolo_tensor att, myolo_tensor out, float one) {
  constexpr int SEG = ET<T>::SEG;
  const int G = feat.c / SEG;
  const int64_t total = (int64_t)feat.n * feat.h * feat.w * G;
  GRID_STRIDE(v, total) {
    int n, y, xx, cg;
    dec(v, G, feat.w, feat.h, n, y, xx, cg);
    float f[SEG], a[SEG];
    Vec<T>::unpack(ldg16(vptr<T>(feat, n, y, xx) + cg * SEG), f);
    Vec<T>::unpack(ldg16(vptr<T>(att, n, 0, 0) + cg * SEG), a);
#pragma unroll
    for (int i = 0; i < SEG; ++i) f[i] = f[i] * a[i] + one * f[i];      // one = 1: FFM (feat*att + feat); 0: ARM / Attention (feat*att)
    stg16(vptr<T>(out, n, y, xx) + cg * SEG, Vec<T>::pack(f));
  }
}
// gfeat (+)= g*(1+att);  gatt[n][c] += sum_pixels g*feat   (block-level partial sums, fp32 atomics)
template <typename T>
__global__ __launch_bounds__(256) void gate_bwd_kernel(myolo_tensor gout, myolo_tensor feat, myolo_tensor att,
                                                       myolo_tensor gfeat, int acc, float* gatt, int G, int PPB,
                                                       int blocks_per_img, float one) {
  constexpr int SEG = ET<T>::SEG;
  extern __shared__ float red[];  // [PPB][G*SEG]
  const int n = blockIdx.x / blocks_per_img, bi = blockIdx.x % blocks_per_img;
  const int cg = threadIdx.x % G, pl = threadIdx.x / G;
  const int HW = feat.h * feat.w;
  float a[SEG], s[SEG];
  Vec<T>::unpack(ldg16(vptr<T>(att, n, 0, 0) + cg * SEG), a);
#pragma unroll
  for (int i = 0; i < SEG; ++i) s[i] = 0.f;
  for (int p = bi * PPB + pl; p < HW; p += blocks_per_img * PPB) {
    const int y = p / feat.w, xx = p - y * feat.w;
    float g[SEG], f[SEG], o[SEG];
    Vec<T>::unpack(ldg16(vptr<T>(gout, n, y, xx) + cg * SEG), g);
    Vec<T>::unpack(ldg16(vptr<T>(feat, n, y, xx) + cg * SEG), f);
#pragma unroll
    for (int i = 0; i < SEG; ++i) { s[i] += g[i] * f[i]; o[i] = g[i] * (one + a[i]); }
    T* gp = vptr<T>(gfeat, n, y, xx) + cg * SEG;
    if (acc) {
      float q[SEG];
      Vec<T>::unpack(ldg16(gp), q);
#pragma unroll
      for (int i = 0; i < SEG; ++i) o[i] += q[i];
    }
    stg16(gp, Vec<T>::pack(o));
  }
#pragma unroll
  for (int i = 0; i < SEG; ++i) red[(size_t)pl * G * SEG + cg * SEG + i] = s[i];
  __syncthreads();
  for (int j = threadIdx.x; j < G * SEG; j += blockDim.x) {
    float t = 0.f;
    for (int q = 0; q < PPB; ++q) t += red[(size_t)q * G * SEG + j];
    atomicAdd(gatt + (size_t)n * G * SEG + j, t);
  }
}

// ---------------------------------------------------------------- add / zero / cast
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(myolo_tensor a, myolo_tensor out, int acc) {
  constexpr int SEG = ET<T>::SEG;
  const int G = out.c / SEG;
  const int64_t total = (int64_t)out.n * out.h * out.w * G;
  GRID_STRIDE(v, total) {
    int n, y, xx, cg;
    dec(v, G, out.w, out.h, n, y, xx, cg);
    uint4 va = ldg16(vptr<T>(a, n, y, xx) + cg * SEG);
    T* op = vptr<T>(out, n, y, xx) + cg * SEG;
    if (acc) {
      float f[SEG], g[SEG];
      Vec<T>::unpack(va, f);
      Vec<T>::unpack(ldg16(op), g);
#pragma unroll
      for (int i = 0; i < SEG; ++i) f[i] += g[i];
      va = Vec<T>::pack(f);
    }
    stg16(op, va);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void zero_kernel(myolo_tensor t) {
  constexpr int SEG = ET<T>::SEG;
  const int G = t.c / SEG;
  const int64_t total = (int64_t)t.n * t.h * t.w * G;
  GRID_STRIDE(v, total) {
    int n, y, xx, cg;
    dec(v, G, t.w, t.h, n, y, xx, cg);
    stg16(vptr<T>(t, n, y, xx) + cg * SEG, uint4{0u, 0u, 0u, 0u});
  }
}
template <typename T>
__global__ __launch_bounds__(256) void cast_from_f32_kernel(const float* __restrict__ src, myolo_tensor out) {
  const int64_t total = (int64_t)out.n * out.h * out.w * out.c;
  GRID_STRIDE(v, total) {
    int64_t r = v;
    const int c = (int)(r % out.c); r /= out.c;
    const int x = (int)(r % out.w); r /= out.w;
    const int y = (int)(r % out.h);
    const int n = (int)(r / out.h);
    vptr<T>(out, n, y, x)[c] = (T)src[v];
  }
}

inline int64_t nvec(const myolo_tensor* t) {
  return (int64_t)t->n * t->h * t->w * (t->c / (t->dtype == MYOLO_F16 ? 8 : 4));
}
inline bool same_nc(const myolo_tensor* a, const myolo_tensor* b) {
  return a->n == b->n && a->c == b->c && a->dtype == b->dtype;
}
inline bool same_shape(const myolo_tensor* a, const myolo_tensor* b) {
  return same_nc(a, b) && a->h == b->h && a->w == b->w;
}

#define DISPATCH(dtype, KERN, grid, block, smem, st, ...)                                        \
  do {                                                                                            \
    if ((dtype) == MYOLO_F16) hipLaunchKernelGGL(KERN<half_t>, dim3(grid), dim3(block), smem, st, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERN<float>, dim3(grid), dim3(block), smem, st, __VA_ARGS__);        \
    MYOLO_CHECK_LAUNCH();                                                                         \
  } while (0)

}  // namespace

extern "C" int myolo_spp_pool_fwd(const myolo_tensor* x, const myolo_tensor* o5, const myolo_tensor* o9,
                                  const myolo_tensor* o13, uint8_t* idx, void* stream) {
  if (!vec_ok(x) || !vec_ok(o5) || !vec_ok(o9) || !vec_ok(o13) || !same_shape(x, o5) || !same_shape(x, o9) ||
      !same_shape(x, o13))
    return MYOLO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int HW = x->h * x->w;
  const int es = x->dtype == MYOLO_F16 ? 2 : 4, seg = 16 / es;
  // tile + 3 row-max planes (16 B/pixel each) + (training only) 3 index planes, for a band of output rows + 12 halo rows; up to 150 KB of
  // the CU's 160 KB.  Bands: as many as it takes to reach ~256 workgroups (batch-1 planes), each at least 4 rows (halo overhead)
  const int blocks = x->n * (x->c / seg);
  int nb = (256 + blocks - 1) / blocks;
  if (nb > x->h / 4) nb = x->h / 4;
  if (nb < 1) nb = 1;
  int band = (x->h + nb - 1) / nb;
  auto need = [&](int rows) { const size_t px = (size_t)(rows + 12) * x->w; return px * 16 * 4 + (idx ? px * 3 * seg : 0); };
  while (need(band) > 150 * 1024 && band > 4) band = (band + 1) / 2;
  nb = (x->h + band - 1) / band;
  const size_t smem = need(band);
  if (smem <= 150 * 1024 && !g_spp_naive) {
    const dim3 grid(blocks, nb);
    if (x->dtype == MYOLO_F16 && !idx && !(g_spp_bwd_form & 8)) {           // eval: packed-half maxima (option spp_bwd_form bit 3: the fp32 plane kernel)
      const size_t smem_h = (size_t)(band + 12) * x->w * 16 * 4;
      auto kern = spp_fwd_plane_h_kernel;
      MYOLO_ENSURE_DYN_SMEM(kern, (int)smem_h);
      hipLaunchKernelGGL(kern, grid, dim3(256), smem_h, st, *x, *o5, *o9, *o13, band);
      MYOLO_CHECK_LAUNCH();
      return 0;
    }
    if (x->dtype == MYOLO_F16) {
      if (idx) { auto kern = spp_fwd_plane_kernel<half_t, true>; MYOLO_ENSURE_DYN_SMEM(kern, (int)smem); hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, *x, *o5, *o9, *o13, idx, band); }
      else { auto kern = spp_fwd_plane_kernel<half_t, false>; MYOLO_ENSURE_DYN_SMEM(kern, (int)smem); hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, *x, *o5, *o9, *o13, idx, band); }
    } else {
      if (idx) { auto kern = spp_fwd_plane_kernel<float, true>; MYOLO_ENSURE_DYN_SMEM(kern, (int)smem); hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, *x, *o5, *o9, *o13, idx, band); }
      else { auto kern = spp_fwd_plane_kernel<float, false>; MYOLO_ENSURE_DYN_SMEM(kern, (int)smem); hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, *x, *o5, *o9, *o13, idx, band); }
    }
    MYOLO_CHECK_LAUNCH();
    return 0;
  }
  const int grid = grid_for(nvec(x), 256, 8192);
  if (x->dtype == MYOLO_F16) {
    if (idx) hipLaunchKernelGGL((spp_fwd_kernel<half_t, true>), dim3(grid), dim3(256), 0, st, *x, *o5, *o9, *o13, idx);
    else hipLaunchKernelGGL((spp_fwd_kernel<half_t, false>), dim3(grid), dim3(256), 0, st, *x, *o5, *o9, *o13, idx);
  } else {
    if (idx) hipLaunchKernelGGL((spp_fwd_kernel<float, true>), dim3(grid), dim3(256), 0, st, *x, *o5, *o9, *o13, idx);
    else hipLaunchKernelGGL((spp_fwd_kernel<float, false>), dim3(grid), dim3(256), 0, st, *x, *o5, *o9, *o13, idx);
  }
  MYOLO_CHECK_LAUNCH();
  return 0;
}
extern "C" int myolo_spp_pool_bwd(const myolo_tensor* g5, const myolo_tensor* g9, const myolo_tensor* g13,
                                  const uint8_t* idx, const myolo_tensor* gx, int accumulate, void* stream) {
  if (!vec_ok(gx) || !vec_ok(g5) || !vec_ok(g9) || !vec_ok(g13) || !idx || !same_shape(gx, g5) ||
      !same_shape(gx, g9) || !same_shape(gx, g13))
    return MYOLO_EINVAL;
  const int HW = gx->h * gx->w;
  const int seg = gx->dtype == MYOLO_F16 ? 8 : 4;
  if (gx->c % 32 == 0 && (size_t)HW * 32 * sizeof(float) <= 144 * 1024 && !g_spp_naive && !(g_spp_bwd_form & 1)) {
    hipStream_t st = (hipStream_t)stream;
    // 16 channels per workgroup when 32 would leave CUs empty (the bs-16 step: 16 images x 8 groups of 32 = 128 workgroups on 256 CUs)
    const bool narrow = (g_spp_bwd_form & 2) ? false : ((g_spp_bwd_form & 4) ? true : gx->n * (gx->c / 32) < 200);
    const int ch = narrow ? 16 : 32;
    const size_t smem = (size_t)HW * ch * sizeof(float);
    const dim3 grid(gx->n * (gx->c / ch));
#define SPP_BWD_CHAN(T_, CH_)                                                                                                   \
    do {                                                                                                                        \
      auto kern = spp_bwd_chan_kernel<T_, CH_, 1024>;                                                                           \
      MYOLO_ENSURE_DYN_SMEM(kern, (int)smem);                                                                                   \
      hipLaunchKernelGGL(kern, grid, dim3(1024), smem, st, *g5, *g9, *g13, idx, *gx, accumulate);                               \
    } while (0)
    if (gx->dtype == MYOLO_F16) { if (narrow) SPP_BWD_CHAN(half_t, 16); else SPP_BWD_CHAN(half_t, 32); }
    else { if (narrow) SPP_BWD_CHAN(float, 16); else SPP_BWD_CHAN(float, 32); }
#undef SPP_BWD_CHAN
    MYOLO_CHECK_LAUNCH();
    return 0;
  }
  if ((size_t)HW * seg * sizeof(float) <= SPP_PLANE_LDS && !g_spp_naive) {
    DISPATCH(gx->dtype, spp_bwd_plane_kernel, gx->n * (gx->c / seg), 256, (size_t)HW * seg * sizeof(float), (hipStream_t)stream,
             *g5, *g9, *g13, idx, *gx, accumulate);
    return 0;
  }
  DISPATCH(gx->dtype, spp_bwd_kernel, grid_for(nvec(gx), 256, 8192), 256, 0, (hipStream_t)stream, *g5, *g9, *g13, idx,
           *gx, accumulate);
  return 0;
}
extern "C" int myolo_copy_up_fwd(const myolo_tensor* x, const myolo_tensor* out, int scale, void* stream) {
  if (!vec_ok(x) || !vec_ok(out) || !same_nc(x, out) || (scale != 1 && scale != 2) || out->h != x->h * scale ||
      out->w != x->w * scale)
    return MYOLO_EINVAL;
  DISPATCH(out->dtype, copy_up_fwd_kernel, grid_for(nvec(out), 256), 256, 0, (hipStream_t)stream, *x, *out,
           scale == 2 ? 1 : 0);
  return 0;
}
extern "C" int myolo_copy_up_bwd(const myolo_tensor* gout, const myolo_tensor* gx, int scale, int accumulate,
                                 void* stream) {
  if (!vec_ok(gx) || !vec_ok(gout) || !same_nc(gx, gout) || (scale != 1 && scale != 2) || gout->h != gx->h * scale ||
      gout->w != gx->w * scale)
    return MYOLO_EINVAL;
  DISPATCH(gx->dtype, copy_up_bwd_kernel, grid_for(nvec(gx), 256), 256, 0, (hipStream_t)stream, *gout, *gx,
           scale == 2 ? 1 : 0, accumulate);
  return 0;
}
static inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }
extern "C" int myolo_bilinear_fwd(const myolo_tensor* x, const myolo_tensor* out, void* stream) {
  if (!vec_ok(x) || !vec_ok(out) || !same_nc(x, out)) return MYOLO_EINVAL;
  DISPATCH(out->dtype, bilinear_fwd_kernel, grid_for(nvec(out), 256), 256, 0, (hipStream_t)stream, *x, *out,
           ac_scale(x->h, out->h), ac_scale(x->w, out->w));
  return 0;
}
extern "C" int myolo_bilinear_bwd(const myolo_tensor* gout, const myolo_tensor* gx, int accumulate, float* scratch, void* stream) {
  if (!vec_ok(gx) || !vec_ok(gout) || !same_nc(gx, gout)) return MYOLO_EINVAL;
  const int64_t foot = (int64_t)(gout->h / (gx->h > 0 ? gx->h : 1)) * (gout->w / (gx->w > 0 ? gx->w : 1));
  const int seg = gx->dtype == MYOLO_F16 ? 8 : 4;
  const int G = gx->c / seg;
  const size_t row_bytes = (size_t)gout->w * gout->c * (gx->dtype == MYOLO_F16 ? 2 : 4);
  // (6x6 already yields 576+ workgroups in the per-pixel kernel and measures faster there)
  if (foot >= 64 && scratch && gx->h * gx->w <= 9 && row_bytes <= 32 * 1024) {
    const int rows_per_iy = gx->h > 1 ? 2 * ((gout->h + gx->h - 2) / (gx->h - 1)) + 6 : gout->h;   // upper bound of one input row's footprint
    int maxrows = (int)(64 * 1024 / row_bytes);
    if (maxrows > BB_ROWS) maxrows = BB_ROWS;
    int per = rows_per_iy * gx->n * gx->h / 512;                     // ~512 workgroups
    if (per < 1) per = 1;
    if (per > maxrows) per = maxrows;
    const int chunks = (rows_per_iy + per - 1) / per;
    hipStream_t st = (hipStream_t)stream;
    if (gx->dtype == MYOLO_F16) {
      hipLaunchKernelGGL(bilinear_bwd_split_kernel<half_t>, dim3(gx->n * gx->h * chunks), dim3(256), per * row_bytes, st, *gout, gx->h,
                         gx->w, ac_scale(gx->h, gout->h), ac_scale(gx->w, gout->w), chunks, per, scratch);
      hipLaunchKernelGGL(bilinear_bwd_finish_kernel<half_t>, dim3(grid_for(nvec(gx), 256)), dim3(256), 0, st, *gx, scratch, accumulate);
    } else {
      hipLaunchKernelGGL(bilinear_bwd_split_kernel<float>, dim3(gx->n * gx->h * chunks), dim3(256), per * row_bytes, st, *gout, gx->h,
                         gx->w, ac_scale(gx->h, gout->h), ac_scale(gx->w, gout->w), chunks, per, scratch);
      hipLaunchKernelGGL(bilinear_bwd_finish_kernel<float>, dim3(grid_for(nvec(gx), 256)), dim3(256), 0, st, *gx, scratch, accumulate);
    }
    MYOLO_CHECK_LAUNCH();
    return 0;
  }
  if (foot >= 64) {   // few input pixels, each gathering thousands of outputs: a workgroup per input pixel
    int gb = 1;
    while (gb < 8 && gb < G) gb <<= 1;
    const int64_t blocks = (int64_t)gx->n * gx->h * gx->w * ((G + gb - 1) / gb);
    if (blocks > 0x7fffffff) return MYOLO_EINVAL;
    DISPATCH(gx->dtype, bilinear_bwd_big_kernel, (int)blocks, 256, 0, (hipStream_t)stream, *gout, *gx,
             ac_scale(gx->h, gout->h), ac_scale(gx->w, gout->w), accumulate, gb);
    return 0;
  }
  DISPATCH(gx->dtype, bilinear_bwd_kernel, grid_for(nvec(gx), 256), 256, 0, (hipStream_t)stream, *gout, *gx,
           ac_scale(gx->h, gout->h), ac_scale(gx->w, gout->w), accumulate);
  return 0;
}
extern "C" int myolo_adaptive_avgpool_fwd(const myolo_tensor* x, const myolo_tensor* out, float* scratch, void* stream) {
  if (!vec_ok(x) || !vec_ok(out) || !same_nc(x, out) || out->h > x->h || out->w > x->w) return MYOLO_EINVAL;
  const int bins = out->n * out->h * out->w;
  const int64_t pix_per_bin = ((int64_t)x->h * x->w) / ((int64_t)out->h * out->w);
  if (scratch && bins < 1024 && pix_per_bin >= 512) {
    int split = (int)(2048 / bins);
    if (split > pix_per_bin / 128) split = (int)(pix_per_bin / 128);
    if (split < 2) split = 2;
    if (x->dtype == MYOLO_F16) {
      hipLaunchKernelGGL(aap_fwd_split_kernel<half_t>, dim3(bins, split), dim3(256), 0, (hipStream_t)stream, *x, out->h, out->w, scratch);
      hipLaunchKernelGGL(aap_finish_kernel<half_t>, dim3(grid_for((int64_t)bins * out->c, 256)), dim3(256), 0, (hipStream_t)stream, *x, *out, scratch);
    } else {
      hipLaunchKernelGGL(aap_fwd_split_kernel<float>, dim3(bins, split), dim3(256), 0, (hipStream_t)stream, *x, out->h, out->w, scratch);
      hipLaunchKernelGGL(aap_finish_kernel<float>, dim3(grid_for((int64_t)bins * out->c, 256)), dim3(256), 0, (hipStream_t)stream, *x, *out, scratch);
    }
    MYOLO_CHECK_LAUNCH();
    return 0;
  }
  DISPATCH(x->dtype, aap_fwd_kernel, bins, 256, 0, (hipStream_t)stream, *x, *out);
  return 0;
}
extern "C" int myolo_adaptive_avgpool_fwd_multi(const myolo_tensor* x, const myolo_tensor* outs, int count, float* scratch, void* stream) {
  if (!x || !outs || !scratch || count < 1 || count > AAP_MAXP || !vec_ok(x)) return MYOLO_EINVAL;
  const int seg = x->dtype == MYOLO_F16 ? 8 : 4;
  const int G = x->c / seg;
  if (G < 1 || G > 256 || 256 % G) return MYOLO_EINVAL;
  const int PL = 256 / G;
  AapFwdMulti mp;
  AapOuts ao;
  mp.np = count; mp.nbins = 0;
  int kmax = 1;
  for (int p = 0; p < AAP_MAXP; ++p) { mp.k[p] = 1; mp.bin0[p] = 0; ao.o[p] = outs[0]; }
  for (int p = 0; p < count; ++p) {
    const myolo_tensor& o = outs[p];
    if (!vec_ok(&o) || !same_nc(x, &o) || o.h != o.w || o.h < 1 || o.h > x->h || o.w > x->w) return MYOLO_EINVAL;
    mp.k[p] = o.h; mp.bin0[p] = mp.nbins; mp.nbins += o.h * o.w; ao.o[p] = o;
    if (o.h > kmax) kmax = o.h;
  }
  // a thread's x range (W/PL pixels, +1 for rounding) must be narrower than a bin minus its one-pixel overlaps: at most 3 bins per range
  if ((x->w + PL - 1) / PL + 2 > x->w / kmax || x->w < PL) return MYOLO_EINVAL;
  const int smem = mp.nbins * x->c * 4;
  if (smem > 60 * 1024) return MYOLO_EINVAL;
  // ~256 workgroups: the measured optimum (round 5, kernel time inside the step / a 2048x1024 frame, profiles/r5k_aap_wgs.txt, r5l_aap_xsplit.txt):
  // 64 / 128 / 256 / 512 workgroups 144 / 78 / 44.7 / 45.2 us for the training map (fewer workgroups: the per-row work serialises; more: the
  // zeroing and flushing of the bin tables, nbins * C words and as many atomics per workgroup, grows with the count); splitting the ROW over
  // workgroups as well 57.7 us at 1024, 98.6 at 2048; one bin table per wave 46.2 -> 48.0 us; four loads in flight per thread neutral
  int rows = (int)(((int64_t)x->h * x->n + g_aap_wgs - 1) / g_aap_wgs);
  if (rows < 1) rows = 1;
  const int gx = (x->h + rows - 1) / rows;
  hipStream_t st = (hipStream_t)stream;
  if (x->dtype == MYOLO_F16) {
    hipLaunchKernelGGL(aap_fwd_multi_kernel<half_t>, dim3(gx, x->n), dim3(256), smem, st, *x, mp, rows, scratch);
    hipLaunchKernelGGL(aap_finish_multi_kernel<half_t>, dim3(grid_for((int64_t)x->n * mp.nbins * x->c, 256)), dim3(256), 0, st, x->h, x->w, x->c,
                       x->n, mp, ao, scratch);
  } else {
    hipLaunchKernelGGL(aap_fwd_multi_kernel<float>, dim3(gx, x->n), dim3(256), smem, st, *x, mp, rows, scratch);
    hipLaunchKernelGGL(aap_finish_multi_kernel<float>, dim3(grid_for((int64_t)x->n * mp.nbins * x->c, 256)), dim3(256), 0, st, x->h, x->w, x->c,
                       x->n, mp, ao, scratch);
  }
  MYOLO_CHECK_LAUNCH();
  return 0;
}
extern "C" int myolo_adaptive_avgpool_bwd(const myolo_tensor* gout, const myolo_tensor* gx, int accumulate,
                                          void* stream) {
  if (!vec_ok(gx) || !vec_ok(gout) || !same_nc(gx, gout)) return MYOLO_EINVAL;
  DISPATCH(gx->dtype, aap_bwd_kernel, grid_for(nvec(gx), 256), 256, 0, (hipStream_t)stream, *gout, *gx, accumulate);
  return 0;
}
extern "C" int myolo_pyramid_upsample_fwd(const myolo_tensor* xs, int count, const myolo_tensor* out, void* stream) {
  if (!xs || !out || count < 1 || count > 4 || !vec_ok(out) || out->c % count) return MYOLO_EINVAL;
  const int seg = out->dtype == MYOLO_F16 ? 8 : 4;
  const int cbr = out->c / count;
  if (cbr % seg) return MYOLO_EINVAL;
  PyrUpF k;
  k.out = *out; k.cgs_per_branch = cbr / seg; k.n = count;
  for (int t = 0; t < 4; ++t) {
    const myolo_tensor& x = xs[t < count ? t : 0];
    if (t < count && (!vec_ok(&x) || x.n != out->n || x.c != cbr || x.dtype != out->dtype)) return MYOLO_EINVAL;
    k.x[t] = x; k.sy[t] = ac_scale(x.h, out->h); k.sx[t] = ac_scale(x.w, out->w);
  }
  DISPATCH(out->dtype, pyr_up_fwd_kernel, grid_for(nvec(out), 256), 256, 0, (hipStream_t)stream, k);
  return 0;
}
extern "C" int myolo_pyramid_upsample_bwd(const myolo_tensor* gout, const myolo_tensor* gxs, int count, const int32_t* accumulate,
                                          float* scratch, void* stream) {
  if (!gout || !gxs || !scratch || count < 1 || count > 4 || !vec_ok(gout) || gout->c % count) return MYOLO_EINVAL;
  const int seg = gout->dtype == MYOLO_F16 ? 8 : 4;
  const int cbr = gout->c / count;
  if (cbr % seg || gout->c / seg > 16) return MYOLO_EINVAL;
  PyrUp k;
  PyrFin f;
  k.gout = *gout; k.cgs_per_branch = cbr / seg; k.nbranch = count; k.rows = 4;
  f.n = count;
  int off = 0;
  for (int t = 0; t < count; ++t) {
    const myolo_tensor& g = gxs[t];
    if (!vec_ok(&g) || g.n != gout->n || g.c != cbr || g.dtype != gout->dtype || g.h != g.w || g.h < 1 || g.h > 6) return MYOLO_EINVAL;
    k.k[t] = g.h; k.sy[t] = ac_scale(g.h, gout->h); k.sx[t] = ac_scale(g.w, gout->w); k.off[t] = off;
    f.gx[t] = g; f.acc[t] = accumulate ? accumulate[t] : 0; f.off[t] = off;
    off += g.n * g.h * g.w * cbr;
  }
  for (int t = count; t < 4; ++t) { k.k[t] = 1; k.sy[t] = k.sx[t] = 0.f; k.off[t] = 0; }
  hipStream_t st = (hipStream_t)stream;
  const int strips = (gout->h + k.rows - 1) / k.rows;
  const int threads = (gout->c / seg) * 16;
  if (gout->dtype == MYOLO_F16) {
    hipLaunchKernelGGL(pyr_up_bwd_kernel<half_t>, dim3(gout->n * strips), dim3(threads), 0, st, k, scratch);
    hipLaunchKernelGGL(pyr_up_bwd_finish_kernel<half_t>, dim3(grid_for(off / seg, 256)), dim3(256), 0, st, f, scratch);
  } else {
    hipLaunchKernelGGL(pyr_up_bwd_kernel<float>, dim3(gout->n * strips), dim3(threads), 0, st, k, scratch);
    hipLaunchKernelGGL(pyr_up_bwd_finish_kernel<float>, dim3(grid_for(off / seg, 256)), dim3(256), 0, st, f, scratch);
  }
  MYOLO_CHECK_LAUNCH();
  return 0;
}
extern "C" int myolo_adaptive_avgpool_bwd_multi(const myolo_tensor* gouts, int count, const myolo_tensor* gx, int accumulate,
                                                void* stream) {
  if (!gouts || count < 1 || count > 4 || !vec_ok(gx)) return MYOLO_EINVAL;
  AapMulti m;
  m.n = count;
  for (int i = 0; i < count; ++i) {
    if (!vec_ok(&gouts[i]) || !same_nc(gx, &gouts[i])) return MYOLO_EINVAL;
    m.g[i] = gouts[i];
  }
  for (int i = 0; i < count; ++i)
    if (gouts[i].h > 8 || gouts[i].w > 8 || gouts[i].h > gx->h || gouts[i].w > gx->w) return MYOLO_EINVAL;   // bin tables: k <= 8
  if (gx->h > 2040 || gx->w > 2040) return MYOLO_EINVAL;
  const size_t smem = (size_t)4 * (gx->h + gx->w) + 128;
  DISPATCH(gx->dtype, aap_bwd_multi_kernel, grid_for(nvec(gx), 256 * 4, 2048), 256, smem, (hipStream_t)stream, m, *gx, accumulate);
  return 0;
}
static int gate_fwd_impl(const myolo_tensor* feat, const myolo_tensor* att, const myolo_tensor* out, float one, void* stream) {
  if (!vec_ok(feat) || !vec_ok(att) || !vec_ok(out) || !same_shape(feat, out) || !same_nc(feat, att) || att->h != 1 ||
      att->w != 1)
    return MYOLO_EINVAL;
  DISPATCH(feat->dtype, gate_fwd_kernel, grid_for(nvec(feat), 256), 256, 0, (hipStream_t)stream, *feat, *att, *out, one);
  return 0;
}
extern "C" int myolo_gate_fwd(const myolo_tensor* feat, const myolo_tensor* att, const myolo_tensor* out, void* stream) {
  return gate_fwd_impl(feat, att, out, 1.0f, stream);
}
extern "C" int myolo_gate_mul_fwd(const myolo_tensor* feat, const myolo_tensor* att, const myolo_tensor* out, void* stream) {
  return gate_fwd_impl(feat, att, out, 0.0f, stream);
}
static int gate_bwd_impl(const myolo_tensor* gout, const myolo_tensor* feat, const myolo_tensor* att,
                         const myolo_tensor* gfeat, int accumulate, float* gatt, float one, void* stream) {
  if (!vec_ok(gout) || !vec_ok(feat) || !vec_ok(att) || !vec_ok(gfeat) || !gatt || !same_shape(gout, feat) ||
      !same_shape(gfeat, feat) || !same_nc(feat, att))
    return MYOLO_EINVAL;
  const int seg = feat->dtype == MYOLO_F16 ? 8 : 4;
  const int G = feat->c / seg;
  if (G > 256) return MYOLO_EINVAL;
  const int PPB = 256 / G;
  const int HW = feat->h * feat->w;
  int bpi = (HW + PPB * 16 - 1) / (PPB * 16);
  if (bpi > 128) bpi = 128;
  if (bpi < 1) bpi = 1;
  const size_t smem = (size_t)PPB * G * seg * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (feat->dtype == MYOLO_F16)
    hipLaunchKernelGGL(gate_bwd_kernel<half_t>, dim3(feat->n * bpi), dim3(G * PPB), smem, st, *gout, *feat, *att, *gfeat,
                       accumulate, gatt, G, PPB, bpi, one);
  else
    hipLaunchKernelGGL(gate_bwd_kernel<float>, dim3(feat->n * bpi), dim3(G * PPB), smem, st, *gout, *feat, *att, *gfeat,
                       accumulate, gatt, G, PPB, bpi, one);
  MYOLO_CHECK_LAUNCH();
  return 0;
}
extern "C" int myolo_gate_bwd(const myolo_tensor* gout, const myolo_tensor* feat, const myolo_tensor* att,
                              const myolo_tensor* gfeat, int accumulate, float* gatt, void* stream) {
  return gate_bwd_impl(gout, feat, att, gfeat, accumulate, gatt, 1.0f, stream);
}
extern "C" int myolo_gate_mul_bwd(const myolo_tensor* gout, const myolo_tensor* feat, const myolo_tensor* att,
                                  const myolo_tensor* gfeat, int accumulate, float* gatt, void* stream) {
  return gate_bwd_impl(gout, feat, att, gfeat, accumulate, gatt, 0.0f, stream);
}
extern "C" int myolo_add(const myolo_tensor* a, const myolo_tensor* out, int accumulate, void* stream) {
  if (!vec_ok(a) || !vec_ok(out) || !same_shape(a, out)) return MYOLO_EINVAL;
  DISPATCH(out->dtype, add_kernel, grid_for(nvec(out), 256), 256, 0, (hipStream_t)stream, *a, *out, accumulate);
  return 0;
}
extern "C" int myolo_fill_zero(const myolo_tensor* t, void* stream) {
  if (!vec_ok(t)) return MYOLO_EINVAL;
  DISPATCH(t->dtype, zero_kernel, grid_for(nvec(t), 256), 256, 0, (hipStream_t)stream, *t);
  return 0;
}
extern "C" int myolo_cast_from_f32(const float* src, const myolo_tensor* out, void* stream) {
  if (!src || !out || !out->ptr) return MYOLO_EINVAL;
  DISPATCH(out->dtype, cast_from_f32_kernel, grid_for((int64_t)out->n * out->h * out->w * out->c, 256), 256, 0,
           (hipStream_t)stream, src, *out);
  return 0;
}
