// Multi-tensor optimizer-side kernels (HBM-bound streaming; one launch per step for ALL parameters instead of the
// reference's ~750 per-tensor launches):
//   mt_sgd          : torch.optim.SGD(momentum, nesterov, weight_decay) step over 3 param groups (train.py:121-137,397)
//                     with the AMP unscale (1/scale) and the found_inf skip folded in
//   mt_check_finite : GradScaler's inf/nan check over every gradient (train.py:397 scaler.step)
//   mt_ema          : ModelEMA.update  v = d*v + (1-d)*p over every floating state_dict tensor (torch_utils.py:290-300)
//   scaler_update   : GradScaler.update growth/backoff bookkeeping on device (no host sync)
// Table format (device, int64 [ntensors][6]): {ptr0, ptr1, ptr2, numel, group, 0}; chunk list (device, int32 [nchunks][2]):
// {tensor index, first element}: workgroup b handles elements [first, first + chunk_elems) of its tensor.
#include "myolo_dev.h"

namespace {

struct Hyper { float lr[8], momentum[8], wd[8]; int nesterov; };

// The kernels stream a chunk with FOUR 16-byte vectors per operand in flight per thread (the first version moved one dword per
// lane and iteration with possibly-aliasing pointers, so every iteration waited out its own loads: 47 / 44 / 40 us for
// 148 / 30 / 89 MB at the serial end of the step).  A chunk whose pointers are not 16-byte aligned (views into a flat buffer at odd
// offsets) takes the dword loop.
__device__ __forceinline__ void sgd_elem(float& pv, float gin, float& b, float inv, float wd, float mom, float lr, int nesterov) {
  float gv = gin * inv;
  if (wd != 0.f) gv += wd * pv;
  b = mom * b + gv;                                           // first step: buf = 0 -> b = g (torch clones the gradient)
  gv = nesterov ? gv + mom * b : b;
  pv = pv - lr * gv;
}

constexpr int MT_UNROLL = 4;
__device__ __forceinline__ bool mt_aligned(const void* a, const void* b, const void* c) {
  return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0;
}

__global__ __launch_bounds__(256) void mt_sgd_kernel(const int64_t* __restrict__ table, const int32_t* __restrict__ chunks, int chunk_elems,
                                                     Hyper h, const float* __restrict__ scale, const float* __restrict__ found_inf) {
  if (found_inf && found_inf[0] != 0.f) return;             // GradScaler.step skips the update on inf/nan gradients
  const int t = chunks[blockIdx.x * 2], start = chunks[blockIdx.x * 2 + 1];
  const int64_t* e = table + (int64_t)t * 6;
  const int64_t n = e[3];
  const int grp = (int)e[4];
  const float lr = h.lr[grp], mom = h.momentum[grp], wd = h.wd[grp];
  const float inv = scale ? 1.f / scale[0] : 1.f;
  int64_t end = (int64_t)start + chunk_elems;
  if (end > n) end = n;
  float* __restrict__ p = reinterpret_cast<float*>(e[0]) + start;
  const float* __restrict__ g = reinterpret_cast<const float*>(e[1]) + start;
  float* __restrict__ buf = reinterpret_cast<float*>(e[2]) + start;
  const int len = (int)(end - start);
  int done = 0;
  if (mt_aligned(p, g, buf)) {
    const int n4 = len >> 2;
    float4* __restrict__ p4 = reinterpret_cast<float4*>(p);
    const float4* __restrict__ g4 = reinterpret_cast<const float4*>(g);
    float4* __restrict__ b4 = reinterpret_cast<float4*>(buf);
    for (int i = threadIdx.x; i < n4; i += 256 * MT_UNROLL) {
      float4 pv[MT_UNROLL], gv[MT_UNROLL], bv[MT_UNROLL];
#pragma unroll
      for (int u = 0; u < MT_UNROLL; ++u)
        if (i + u * 256 < n4) { pv[u] = p4[i + u * 256]; gv[u] = g4[i + u * 256]; bv[u] = b4[i + u * 256]; }
#pragma unroll
      for (int u = 0; u < MT_UNROLL; ++u)
        if (i + u * 256 < n4) {
          sgd_elem(pv[u].x, gv[u].x, bv[u].x, inv, wd, mom, lr, h.nesterov);
          sgd_elem(pv[u].y, gv[u].y, bv[u].y, inv, wd, mom, lr, h.nesterov);
          sgd_elem(pv[u].z, gv[u].z, bv[u].z, inv, wd, mom, lr, h.nesterov);
          sgd_elem(pv[u].w, gv[u].w, bv[u].w, inv, wd, mom, lr, h.nesterov);
          b4[i + u * 256] = bv[u];
          p4[i + u * 256] = pv[u];
        }
    }
    done = n4 << 2;
  }
  for (int i = done + threadIdx.x; i < len; i += 256) {
    float pv = p[i], b = buf[i];
    sgd_elem(pv, g[i], b, inv, wd, mom, lr, h.nesterov);
    buf[i] = b;
    p[i] = pv;
  }
}

__global__ __launch_bounds__(256) void mt_check_kernel(const int64_t* __restrict__ table, const int32_t* __restrict__ chunks, int chunk_elems,
                                                       int which, float* found_inf) {
  const int t = chunks[blockIdx.x * 2], start = chunks[blockIdx.x * 2 + 1];
  const int64_t* e = table + (int64_t)t * 6;
  const int64_t n = e[3];
  int64_t end = (int64_t)start + chunk_elems;
  if (end > n) end = n;
  const float* __restrict__ g = reinterpret_cast<const float*>(e[which]) + start;
  const int len = (int)(end - start);
  bool bad = false;
  int done = 0;
  auto chk = [&](float v) { bad |= !(fabsf(v) <= 3.402823466e+38f); };     // inf or nan
  if (mt_aligned(g, nullptr, nullptr)) {
    const int n4 = len >> 2;
    const float4* __restrict__ g4 = reinterpret_cast<const float4*>(g);
    for (int i = threadIdx.x; i < n4; i += 256 * MT_UNROLL) {
      float4 gv[MT_UNROLL];
#pragma unroll
      for (int u = 0; u < MT_UNROLL; ++u) gv[u] = i + u * 256 < n4 ? g4[i + u * 256] : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < MT_UNROLL; ++u) { chk(gv[u].x); chk(gv[u].y); chk(gv[u].z); chk(gv[u].w); }
    }
    done = n4 << 2;
  }
  for (int i = done + threadIdx.x; i < len; i += 256) chk(g[i]);
  if (__any(bad) && (threadIdx.x & 63) == 0) found_inf[0] = 1.f;
}

__global__ __launch_bounds__(256) void mt_ema_kernel(const int64_t* __restrict__ table, const int32_t* __restrict__ chunks, int chunk_elems,
                                                     float d) {
  const int t = chunks[blockIdx.x * 2], start = chunks[blockIdx.x * 2 + 1];
  const int64_t* e = table + (int64_t)t * 6;
  const int64_t n = e[3];
  int64_t end = (int64_t)start + chunk_elems;
  if (end > n) end = n;
  float* __restrict__ v = reinterpret_cast<float*>(e[0]) + start;
  const float* __restrict__ m = reinterpret_cast<const float*>(e[1]) + start;
  const int len = (int)(end - start);
  const float omd = 1.f - d;
  auto ema = [&](float vv, float mm) { float x = vv * d; x += omd * mm; return x; };   // torch_utils.py:298-299: v *= d; v += (1-d)*m
  int done = 0;
  if (mt_aligned(v, m, nullptr)) {
    const int n4 = len >> 2;
    float4* __restrict__ v4 = reinterpret_cast<float4*>(v);
    const float4* __restrict__ m4 = reinterpret_cast<const float4*>(m);
    for (int i = threadIdx.x; i < n4; i += 256 * MT_UNROLL) {
      float4 vv[MT_UNROLL], mm[MT_UNROLL];
#pragma unroll
      for (int u = 0; u < MT_UNROLL; ++u)
        if (i + u * 256 < n4) { vv[u] = v4[i + u * 256]; mm[u] = m4[i + u * 256]; }
#pragma unroll
      for (int u = 0; u < MT_UNROLL; ++u)
        if (i + u * 256 < n4)
          v4[i + u * 256] = float4{ema(vv[u].x, mm[u].x), ema(vv[u].y, mm[u].y), ema(vv[u].z, mm[u].z), ema(vv[u].w, mm[u].w)};
    }
    done = n4 << 2;
  }
  for (int i = done + threadIdx.x; i < len; i += 256) v[i] = ema(v[i], m[i]);
}

__global__ void scaler_update_kernel(float* scale, int32_t* tracker, float* found_inf, float growth, float backoff, int interval) {
  if (found_inf[0] != 0.f) { scale[0] *= backoff; tracker[0] = 0; }
  else if (++tracker[0] == interval) { scale[0] *= growth; tracker[0] = 0; }
  found_inf[0] = 0.f;
}

}  // namespace

extern "C" int myolo_mt_sgd(const int64_t* table, const int32_t* chunks, int nchunks, int chunk_elems, const myolo_sgd_hyper* hy,
                            const float* scale, const float* found_inf, void* stream) {
  if (!table || !chunks || !hy || nchunks < 0 || chunk_elems < 1) return MYOLO_EINVAL;
  if (nchunks == 0) return 0;
  Hyper h;
  for (int i = 0; i < 8; ++i) { h.lr[i] = hy->lr[i]; h.momentum[i] = hy->momentum[i]; h.wd[i] = hy->weight_decay[i]; }
  h.nesterov = hy->nesterov;
  hipLaunchKernelGGL(mt_sgd_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, table, chunks, chunk_elems, h, scale, found_inf);
  MYOLO_CHECK_LAUNCH();
  return 0;
}
extern "C" int myolo_mt_check_finite(const int64_t* table, const int32_t* chunks, int nchunks, int chunk_elems, int which,
                                     float* found_inf, void* stream) {
  if (!table || !chunks || !found_inf || nchunks < 0 || chunk_elems < 1 || which < 0 || which > 2) return MYOLO_EINVAL;
  if (nchunks == 0) return 0;
  hipLaunchKernelGGL(mt_check_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, table, chunks, chunk_elems, which, found_inf);
  MYOLO_CHECK_LAUNCH();
  return 0;
}
extern "C" int myolo_mt_ema(const int64_t* table, const int32_t* chunks, int nchunks, int chunk_elems, float decay, void* stream) {
  if (!table || !chunks || nchunks < 0 || chunk_elems < 1) return MYOLO_EINVAL;
  if (nchunks == 0) return 0;
  hipLaunchKernelGGL(mt_ema_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, table, chunks, chunk_elems, decay);
  MYOLO_CHECK_LAUNCH();
  return 0;
}
extern "C" int myolo_scaler_update(float* scale, int32_t* growth_tracker, float* found_inf, float growth, float backoff,
                                   int interval, void* stream) {
  if (!scale || !growth_tracker || !found_inf) return MYOLO_EINVAL;
  hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, scale, growth_tracker, found_inf, growth,
                     backoff, interval);
  MYOLO_CHECK_LAUNCH();
  return 0;
}
