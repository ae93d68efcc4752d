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

__global__ __launch_bounds__(256) void mt_sgd_kernel(const int64_t* table, const int32_t* chunks, int chunk_elems, Hyper h,
                                                     const float* scale, const float* found_inf) {
  if (found_inf && found_inf[0] != 0.f) return;             // GradScaler.step skips the update on inf/nan gradients
  const int t = chunks[blockIdx.x * 2], start = chunks[blockIdx.x * 2 + 1];
  const int64_t* e = table + (int64_t)t * 6;
  float* p = reinterpret_cast<float*>(e[0]);
  const float* g = reinterpret_cast<const float*>(e[1]);
  float* buf = reinterpret_cast<float*>(e[2]);
  const int64_t n = e[3];
  const int grp = (int)e[4];
  const float lr = h.lr[grp], mom = h.momentum[grp], wd = h.wd[grp];
  const float inv = scale ? 1.f / scale[0] : 1.f;
  int64_t end = (int64_t)start + chunk_elems;
  if (end > n) end = n;
  for (int64_t i = (int64_t)start + threadIdx.x; i < end; i += 256) {
    const float pv = p[i];
    float gv = g[i] * inv;
    if (wd != 0.f) gv += wd * pv;
    const float b = mom * buf[i] + gv;                        // first step: buf = 0 -> b = g (torch clones the gradient)
    buf[i] = b;
    gv = h.nesterov ? gv + mom * b : b;
    p[i] = pv - lr * gv;
  }
}

__global__ __launch_bounds__(256) void mt_check_kernel(const int64_t* table, const int32_t* chunks, int chunk_elems, int which,
                                                       float* found_inf) {
  const int t = chunks[blockIdx.x * 2], start = chunks[blockIdx.x * 2 + 1];
  const int64_t* e = table + (int64_t)t * 6;
  const float* g = reinterpret_cast<const float*>(e[which]);
  const int64_t n = e[3];
  int64_t end = (int64_t)start + chunk_elems;
  if (end > n) end = n;
  bool bad = false;
  for (int64_t i = (int64_t)start + threadIdx.x; i < end; i += 256) {
    const float v = g[i];
    bad |= !(fabsf(v) <= 3.402823466e+38f);                  // inf or nan
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) found_inf[0] = 1.f;
}

__global__ __launch_bounds__(256) void mt_ema_kernel(const int64_t* table, const int32_t* chunks, int chunk_elems, float d) {
  const int t = chunks[blockIdx.x * 2], start = chunks[blockIdx.x * 2 + 1];
  const int64_t* e = table + (int64_t)t * 6;
  float* v = reinterpret_cast<float*>(e[0]);
  const float* m = reinterpret_cast<const float*>(e[1]);
  const int64_t n = e[3];
  int64_t end = (int64_t)start + chunk_elems;
  if (end > n) end = n;
  const float omd = 1.f - d;
  for (int64_t i = (int64_t)start + threadIdx.x; i < end; i += 256) {
    float x = v[i] * d;                                        // torch_utils.py:298-299: v *= d; v += (1-d)*m
    x += omd * m[i];
    v[i] = x;
  }
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
