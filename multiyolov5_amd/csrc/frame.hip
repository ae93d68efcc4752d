// detect.py's per-frame host work moved next to the network: after the model itself runs in ~2.5 ms the
// reference's numpy / cv2 steps around it (datasets.py:818-848 letterbox border + `img[:, :, ::-1].transpose(2, 0, 1)` (185),
// detect.py:135-137 uint8 -> half -> /255, detect.py:193-194 label2image + cv2.addWeighted) and their PCIe round trips dominate a
// frame.  Both kernels are pure byte shuffles, HBM-bound: one pass, 16-byte stores where the layout allows.
#include "myolo_dev.h"

namespace {

// model input [1,3,H,W] (NCHW, fp16|fp32) from a uint8 HWC frame placed at (top,left) inside a constant border:
//   out[c][y][x] = lut[ frame[y-top][x-left][swap ? 2-c : c] ]   inside,   lut[pad]   on the border.
// lut = the 256 values of `torch.arange(256).to(dtype) / 255.0` computed by torch itself (utils/datasets_dev.py): the normalisation
// is bit-identical to detect.py:136-137 in either dtype by construction.
template <typename D>
__global__ __launch_bounds__(256) void frame_pack_kernel(const uint8_t* __restrict__ im, int h0, int w0, int swap, int top, int left,
                                                         int H, int W, int pad, D* __restrict__ out, const D* __restrict__ lut_g) {
  __shared__ D lut[256];
  lut[threadIdx.x] = lut_g[threadIdx.x];
  __syncthreads();
  const int64_t total = (int64_t)H * W;
  const D padv = lut[pad & 255];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
    const int sy = y - top, sx = x - left;
    D v0 = padv, v1 = padv, v2 = padv;
    if ((unsigned)sy < (unsigned)h0 && (unsigned)sx < (unsigned)w0) {
      const uint8_t* p = im + ((int64_t)sy * w0 + sx) * 3;
      const uint8_t a = p[0], b = p[1], c = p[2];
      v0 = lut[swap ? c : a]; v1 = lut[b]; v2 = lut[swap ? a : c];
    }
    out[i] = v0; out[total + i] = v1; out[2 * total + i] = v2;
  }
}

// the same with the frame RESAMPLED to (rh, rw) first: cv2.resize(img, new_unpad, interpolation=cv2.INTER_LINEAR) of letterbox
// (datasets.py:843-844) for 8-bit images, restated from OpenCV's resize.cpp (cv2 is absent here: "parity unpinned"):
//   fx = (float)((dx + 0.5) * (w0 / (double) rw) - 0.5); sx = floor(fx); fx -= sx; left / right border: fx = 0, sx clamped;
//   coefficients rounded (half to even) to 1/2048 fixed point (INTER_RESIZE_COEF_SCALE), horizontal pass in int32,
//   vertical pass ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2, source rows clamped;
//   an exact 2x down-scale in both directions is INTER_AREA's (a + b + c + d + 2) >> 2 (resize(): "INTER_AREA (fast) also is
//   equal to INTER_LINEAR" branch) -- the 2048x1024 -> 1024x512 case of the detect.py benchmark.
__device__ __forceinline__ int cv_coef(float v) { return (int)rintf(v * 2048.f); }

template <typename D>
__global__ __launch_bounds__(256) void frame_resize_pack_kernel(const uint8_t* __restrict__ im, int h0, int w0, int rh, int rw, int swap,
                                                                int top, int left, int H, int W, int pad, D* __restrict__ out,
                                                                const D* __restrict__ lut_g, double scale_x, double scale_y, int area2) {
  __shared__ D lut[256];
  lut[threadIdx.x] = lut_g[threadIdx.x];
  __syncthreads();
  const int64_t total = (int64_t)H * W;
  const D padv = lut[pad & 255];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
    const int dy = y - top, dx = x - left;
    D v[3] = {padv, padv, padv};
    if ((unsigned)dy < (unsigned)rh && (unsigned)dx < (unsigned)rw) {
      int c3[3];
      if (area2) {
        const uint8_t* p0 = im + ((int64_t)(2 * dy) * w0 + 2 * dx) * 3;
        const uint8_t* p1 = p0 + (int64_t)w0 * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) c3[c] = (p0[c] + p0[3 + c] + p1[c] + p1[3 + c] + 2) >> 2;
      } else {
        float fx = (float)(((double)dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= w0 - 1) { fx = 0.f; sx = w0 - 1; }
        float fy = (float)(((double)dy + 0.5) * scale_y - 0.5);
        const int sy = (int)floorf(fy);
        fy -= (float)sy;
        const int a0 = cv_coef(1.f - fx), a1 = cv_coef(fx), b0 = cv_coef(1.f - fy), b1 = cv_coef(fy);
        const int x1 = sx + 1 < w0 ? sx + 1 : w0 - 1;
        const int y0 = sy < 0 ? 0 : (sy > h0 - 1 ? h0 - 1 : sy), y1 = sy + 1 < 0 ? 0 : (sy + 1 > h0 - 1 ? h0 - 1 : sy + 1);
        const uint8_t* r0 = im + (int64_t)y0 * w0 * 3;
        const uint8_t* r1 = im + (int64_t)y1 * w0 * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int h0v = r0[sx * 3 + c] * a0 + r0[x1 * 3 + c] * a1;
          const int h1v = r1[sx * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
          int t = (((b0 * (h0v >> 4)) >> 16) + ((b1 * (h1v >> 4)) >> 16) + 2) >> 2;
          c3[c] = t < 0 ? 0 : (t > 255 ? 255 : t);
        }
      }
      v[0] = lut[swap ? c3[2] : c3[0]]; v[1] = lut[c3[1]]; v[2] = lut[swap ? c3[0] : c3[2]];
    }
    out[i] = v[0]; out[total + i] = v[1]; out[2 * total + i] = v[2];
  }
}

// mask[y][x][:] = colormap[label][reversed if swap]  (label2image(...)[:, :, ::-1]);  dst = saturate(rint(mask*alpha + im0*beta + gamma))
// = cv2.addWeighted on CV_8U (float arithmetic, round half to even).  Either output may be NULL.
template <typename LT>
__global__ __launch_bounds__(256) void seg_blend_kernel(const LT* __restrict__ labels, const uint8_t* __restrict__ im0, int64_t total,
                                                        const uint8_t* __restrict__ cmap, int ncls, int swap, float alpha, float beta,
                                                        float gamma, uint8_t* __restrict__ mask, uint8_t* __restrict__ dst) {
  __shared__ uint8_t cm[256 * 3];
  for (int i = threadIdx.x; i < ncls * 3; i += 256) cm[i] = cmap[i];
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int l = (int)labels[i];
    l = l < 0 ? 0 : (l >= ncls ? ncls - 1 : l);
    const uint8_t r = cm[l * 3], g = cm[l * 3 + 1], b = cm[l * 3 + 2];
    const uint8_t m0 = swap ? b : r, m1 = g, m2 = swap ? r : b;
    if (mask) { mask[i * 3] = m0; mask[i * 3 + 1] = m1; mask[i * 3 + 2] = m2; }
    if (dst) {
      const uint8_t* p = im0 + i * 3;
      // separately rounded products and sums (no fma contraction): the order a float32 numpy restatement evaluates
      const float t0 = __fadd_rn(__fadd_rn(__fmul_rn((float)m0, alpha), __fmul_rn((float)p[0], beta)), gamma);
      const float t1 = __fadd_rn(__fadd_rn(__fmul_rn((float)m1, alpha), __fmul_rn((float)p[1], beta)), gamma);
      const float t2 = __fadd_rn(__fadd_rn(__fmul_rn((float)m2, alpha), __fmul_rn((float)p[2], beta)), gamma);
      dst[i * 3] = (uint8_t)fminf(fmaxf(rintf(t0), 0.f), 255.f);
      dst[i * 3 + 1] = (uint8_t)fminf(fmaxf(rintf(t1), 0.f), 255.f);
      dst[i * 3 + 2] = (uint8_t)fminf(fmaxf(rintf(t2), 0.f), 255.f);
    }
  }
}

}  // namespace

extern "C" int myolo_frame_pack(const uint8_t* frame_hwc, int h0, int w0, int swap_rb, int top, int left, int H, int W, int pad_value,
                                void* out_nchw, int out_dtype, const void* lut256, void* stream) {
  if (!frame_hwc || !out_nchw || !lut256 || h0 < 1 || w0 < 1 || H < 1 || W < 1 || top < 0 || left < 0 || top + h0 > H || left + w0 > W ||
      (out_dtype != MYOLO_F16 && out_dtype != MYOLO_F32))
    return MYOLO_EINVAL;
  const int grid = grid_for((int64_t)H * W, 256, 4096);
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == MYOLO_F16)
    hipLaunchKernelGGL(frame_pack_kernel<half_t>, dim3(grid), dim3(256), 0, st, frame_hwc, h0, w0, swap_rb, top, left, H, W, pad_value,
                       (half_t*)out_nchw, (const half_t*)lut256);
  else
    hipLaunchKernelGGL(frame_pack_kernel<float>, dim3(grid), dim3(256), 0, st, frame_hwc, h0, w0, swap_rb, top, left, H, W, pad_value,
                       (float*)out_nchw, (const float*)lut256);
  MYOLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int myolo_frame_resize_pack(const uint8_t* frame_hwc, int h0, int w0, int rh, int rw, int swap_rb, int top, int left, int H,
                                       int W, int pad_value, void* out_nchw, int out_dtype, const void* lut256, void* stream) {
  if (!frame_hwc || !out_nchw || !lut256 || h0 < 1 || w0 < 1 || rh < 1 || rw < 1 || H < 1 || W < 1 || top < 0 || left < 0 ||
      top + rh > H || left + rw > W || (out_dtype != MYOLO_F16 && out_dtype != MYOLO_F32))
    return MYOLO_EINVAL;
  const int grid = grid_for((int64_t)H * W, 256, 4096);
  hipStream_t st = (hipStream_t)stream;
  const double sx = (double)w0 / rw, sy = (double)h0 / rh;
  const int area2 = (w0 == 2 * rw && h0 == 2 * rh) ? 1 : 0;
  if (out_dtype == MYOLO_F16)
    hipLaunchKernelGGL(frame_resize_pack_kernel<half_t>, dim3(grid), dim3(256), 0, st, frame_hwc, h0, w0, rh, rw, swap_rb, top, left, H, W,
                       pad_value, (half_t*)out_nchw, (const half_t*)lut256, sx, sy, area2);
  else
    hipLaunchKernelGGL(frame_resize_pack_kernel<float>, dim3(grid), dim3(256), 0, st, frame_hwc, h0, w0, rh, rw, swap_rb, top, left, H, W,
                       pad_value, (float*)out_nchw, (const float*)lut256, sx, sy, area2);
  MYOLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int myolo_seg_blend(const void* labels, int label_dtype, const uint8_t* im0_hwc, int h, int w, const uint8_t* colormap_rgb,
                               int ncls, int swap_rb, float alpha, float beta, float gamma, uint8_t* mask_hwc, uint8_t* dst_hwc,
                               void* stream) {
  if (!labels || !colormap_rgb || ncls < 1 || ncls > 256 || h < 1 || w < 1 || (!mask_hwc && !dst_hwc) || (dst_hwc && !im0_hwc) ||
      (label_dtype != MYOLO_U8 && label_dtype != MYOLO_I64))
    return MYOLO_EINVAL;
  const int64_t total = (int64_t)h * w;
  const int grid = grid_for(total, 256, 4096);
  hipStream_t st = (hipStream_t)stream;
  if (label_dtype == MYOLO_U8)
    hipLaunchKernelGGL(seg_blend_kernel<uint8_t>, dim3(grid), dim3(256), 0, st, (const uint8_t*)labels, im0_hwc, total, colormap_rgb, ncls,
                       swap_rb, alpha, beta, gamma, mask_hwc, dst_hwc);
  else
    hipLaunchKernelGGL(seg_blend_kernel<int64_t>, dim3(grid), dim3(256), 0, st, (const int64_t*)labels, im0_hwc, total, colormap_rgb, ncls,
                       swap_rb, alpha, beta, gamma, mask_hwc, dst_hwc);
  MYOLO_CHECK_LAUNCH();
  return 0;
}
