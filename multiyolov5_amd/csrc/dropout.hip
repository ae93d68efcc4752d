// Train-mode nn.Dropout of the Base / BiSe heads (reference models/yolo.py:65,140): counter-based hash RNG
// (one 64-bit draw counter in device memory, advanced by its own tiny launch so a captured hipGraph produces a fresh
// mask on every replay), keep-mask stored as u8 for the backward.  Statistical parity only (SURVEY §8c dropout caveat).
#include "myolo_dev.h"

namespace {

__device__ __forceinline__ uint32_t mix(uint64_t seed, uint64_t idx) {
  uint64_t z = seed * 0x9E3779B97F4A7C15ull + idx + 0x632BE59BD9B4E019ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)((z ^ (z >> 31)) >> 32);
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_fwd_kernel(myolo_tensor x, myolo_tensor out, uint8_t* mask, float p,
                                                          const uint64_t* counter) {
  constexpr int SEG = ET<T>::SEG;
  const uint64_t seed = counter[0];
  const int G = x.c / SEG;
  const int64_t total = (int64_t)x.n * x.h * x.w * G;
  const float scale = 1.f / (1.f - p);
  const uint32_t thr = (uint32_t)(p * 4294967296.0);
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = v;
    const int cg = (int)(r % G); r /= G;
    const int xx = (int)(r % x.w); r /= x.w;
    const int y = (int)(r % x.h); const int n = (int)(r / x.h);
    float f[SEG];
    Vec<T>::unpack(ldg16(vptr<T>(x, n, y, xx) + cg * SEG), f);
#pragma unroll
    for (int i = 0; i < SEG; ++i) {
      const bool keep = mix(seed, (uint64_t)v * SEG + i) >= thr;
      mask[v * SEG + i] = keep;
      f[i] = keep ? f[i] * scale : 0.f;
    }
    stg16(vptr<T>(out, n, y, xx) + cg * SEG, Vec<T>::pack(f));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_bwd_kernel(myolo_tensor gout, const uint8_t* mask, myolo_tensor gx, float p, int acc) {
  constexpr int SEG = ET<T>::SEG;
  const int G = gx.c / SEG;
  const int64_t total = (int64_t)gx.n * gx.h * gx.w * G;
  const float scale = 1.f / (1.f - p);
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = v;
    const int cg = (int)(r % G); r /= G;
    const int xx = (int)(r % gx.w); r /= gx.w;
    const int y = (int)(r % gx.h); const int n = (int)(r / gx.h);
    float f[SEG];
    Vec<T>::unpack(ldg16(vptr<T>(gout, n, y, xx) + cg * SEG), f);
#pragma unroll
    for (int i = 0; i < SEG; ++i) f[i] = mask[v * SEG + i] ? f[i] * scale : 0.f;
    T* gp = vptr<T>(gx, n, y, xx) + cg * SEG;
    if (acc) {
      float o[SEG];
      Vec<T>::unpack(ldg16(gp), o);
#pragma unroll
      for (int i = 0; i < SEG; ++i) f[i] += o[i];
    }
    stg16(gp, Vec<T>::pack(f));
  }
}

__global__ void rng_advance_kernel(uint64_t* counter) { counter[0] += 1; }

}  // namespace

extern "C" int myolo_dropout_fwd(const myolo_tensor* x, const myolo_tensor* out, uint8_t* mask, float p,
                                 uint64_t* counter, void* stream) {
  if (!x || !out || !x->ptr || !out->ptr || !mask || !counter || p < 0.f || p >= 1.f) return MYOLO_EINVAL;
  const int seg = x->dtype == MYOLO_F16 ? 8 : 4;
  if (x->c % seg || out->c != x->c) return MYOLO_EINVAL;
  const int grid = grid_for((int64_t)x->n * x->h * x->w * (x->c / seg), 256);
  hipStream_t st = (hipStream_t)stream;
  if (x->dtype == MYOLO_F16) hipLaunchKernelGGL(dropout_fwd_kernel<half_t>, dim3(grid), dim3(256), 0, st, *x, *out, mask, p, counter);
  else hipLaunchKernelGGL(dropout_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, *x, *out, mask, p, counter);
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, st, counter);
  MYOLO_CHECK_LAUNCH();
  return 0;
}
extern "C" int myolo_dropout_bwd(const myolo_tensor* gout, const uint8_t* mask, const myolo_tensor* gx, float p,
                                 int accumulate, void* stream) {
  if (!gout || !gx || !gout->ptr || !gx->ptr || !mask) return MYOLO_EINVAL;
  const int seg = gx->dtype == MYOLO_F16 ? 8 : 4;
  const int grid = grid_for((int64_t)gx->n * gx->h * gx->w * (gx->c / seg), 256);
  hipStream_t st = (hipStream_t)stream;
  if (gx->dtype == MYOLO_F16) hipLaunchKernelGGL(dropout_bwd_kernel<half_t>, dim3(grid), dim3(256), 0, st, *gout, mask, *gx, p, accumulate);
  else hipLaunchKernelGGL(dropout_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, *gout, mask, *gx, p, accumulate);
  MYOLO_CHECK_LAUNCH();
  return 0;
}
