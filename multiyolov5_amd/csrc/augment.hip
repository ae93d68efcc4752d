// Training-time augmentation on the device: with the network at > 1600 img/s per GPU the reference's PIL / cv2
// workers (SegmentationDataset.py:118-151 `_sync_transform`, utils/datasets.py:646-724,851-937 mosaic / random_perspective /
// augment_hsv) cannot feed one MI355X.  The random draws stay on the host (utils/augment.py repeats the reference's `random` call
// sequence); these kernels move the pixels.  All of it is byte / integer work, HBM-bound, one pass per sample.
//
//   seg_sync       : mirror -> PIL resize (BILINEAR for the image, NEAREST for the label map) -> pad (0 / 255) -> crop, fused: only the
//                    pixels of the crop are computed.  Pillow's 8-bit resampler restated exactly (Resample.c): separable triangle
//                    filter widened by the down-scale factor, coefficients in 2^-22 fixed point, the horizontal pass rounded to
//                    uint8 before the vertical one.  The coefficient / bound tables are built on the host in double like
//                    precompute_coeffs() does, so the device side is pure integer arithmetic (bit-exact against PIL).
//                    Labels go through the dataset's id -> train-id table (CitySegmentation._class_to_index) to int64.
//   color_jitter   : torchvision ColorJitter on a PIL image = ImageEnhance brightness / contrast / color (Image.blend with a
//                    degenerate image) + the HSV hue shift, in a caller-given order, then ToTensor (uint8 -> float / 255, CHW).
//   det_mosaic_warp: 4-image mosaic canvas + cv2.warpAffine (INTER_LINEAR, border 114) + augment_hsv + left-right flip + BGR->RGB CHW,
//                    fused: every output pixel is traced back through the affine map into the source images (no canvas in memory).
#include "myolo_dev.h"

// every float / double expression below restates C code compiled for baseline x86-64 (Pillow, OpenCV): no fused multiply-add --
// this file is compiled with -ffp-contract=off (multiyolov5_amd/build.py PER_FILE_FLAGS; hipcc's default "fast" ignores pragmas)

namespace {

constexpr int PIL_BITS = 22;              // PRECISION_BITS = 32 - 8 - 2 (Resample.c)

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

struct SegSync {
  const uint8_t* img; const uint8_t* mask; int H0, W0, flip;
  int ow, oh;                              // size after the random rescale
  const int* hb; const int* hk; int ksh;   // horizontal pass: bounds [ow][2] = (first source column, count), coefficients [ow][ksh]
  const int* vb; const int* vk; int ksv;   // vertical pass
  const int* xin; const int* yin;          // NEAREST source column / row of every resized column / row
  int x1, y1, wc, hc;                      // crop window inside the (padded) resized image
  uint8_t* out_img; int64_t* out_lab; const int64_t* lab_lut;
};

__global__ __launch_bounds__(256) void seg_sync_kernel(const SegSync p) {
  const int64_t total = (int64_t)p.hc * p.wc;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cy = (int)(i / p.wc), cx = (int)(i - (int64_t)cy * p.wc);
    const int ry = p.y1 + cy, rx = p.x1 + cx;
    int o0 = 0, o1 = 0, o2 = 0, lab = 255;                   // ImageOps.expand fill: 0 for the image, 255 for the label map
    if (rx < p.ow && ry < p.oh) {
      const int xmin = p.hb[2 * rx], xn = p.hb[2 * rx + 1];
      const int ymin = p.vb[2 * ry], yn = p.vb[2 * ry + 1];
      const int* hk = p.hk + (int64_t)rx * p.ksh;
      const int* vk = p.vk + (int64_t)ry * p.ksv;
      int a0 = 1 << (PIL_BITS - 1), a1 = a0, a2 = a0;
      for (int j = 0; j < yn; ++j) {
        const uint8_t* row = p.img + (int64_t)(ymin + j) * p.W0 * 3;
        int h0 = 1 << (PIL_BITS - 1), h1 = h0, h2 = h0;
        for (int k = 0; k < xn; ++k) {
          const int sx = p.flip ? p.W0 - 1 - (xmin + k) : xmin + k;
          const uint8_t* px = row + sx * 3;
          const int c = hk[k];
          h0 += px[0] * c; h1 += px[1] * c; h2 += px[2] * c;
        }
        const int w = vk[j];                                   // the horizontal pass is stored as uint8 before the vertical one
        a0 += clip8(h0 >> PIL_BITS) * w; a1 += clip8(h1 >> PIL_BITS) * w; a2 += clip8(h2 >> PIL_BITS) * w;
      }
      o0 = clip8(a0 >> PIL_BITS); o1 = clip8(a1 >> PIL_BITS); o2 = clip8(a2 >> PIL_BITS);
      if (p.mask) {
        const int sx = p.flip ? p.W0 - 1 - p.xin[rx] : p.xin[rx];
        lab = p.mask[(int64_t)p.yin[ry] * p.W0 + sx];
      }
    }
    if (p.out_img) { uint8_t* o = p.out_img + i * 3; o[0] = (uint8_t)o0; o[1] = (uint8_t)o1; o[2] = (uint8_t)o2; }
    if (p.out_lab) p.out_lab[i] = p.lab_lut[lab];
  }
}

}  // namespace

extern "C" int myolo_seg_sync_transform(const myolo_seg_sync_desc* d, void* stream) {
  if (!d || !d->img || d->H0 < 1 || d->W0 < 1 || d->ow < 1 || d->oh < 1 || d->wc < 1 || d->hc < 1) return MYOLO_EINVAL;
  if (!d->hb || !d->hk || !d->vb || !d->vk || d->ksh < 1 || d->ksv < 1) return MYOLO_EINVAL;
  if (d->x1 < 0 || d->y1 < 0) return MYOLO_EINVAL;
  if (d->out_lab && (!d->mask || !d->xin || !d->yin || !d->lab_lut)) return MYOLO_EINVAL;
  if (!d->out_img && !d->out_lab) return MYOLO_EINVAL;
  SegSync k;
  k.img = d->img; k.mask = d->out_lab ? d->mask : nullptr; k.H0 = d->H0; k.W0 = d->W0; k.flip = d->flip;
  k.ow = d->ow; k.oh = d->oh; k.hb = d->hb; k.hk = d->hk; k.ksh = d->ksh; k.vb = d->vb; k.vk = d->vk; k.ksv = d->ksv;
  k.xin = d->xin; k.yin = d->yin; k.x1 = d->x1; k.y1 = d->y1; k.wc = d->wc; k.hc = d->hc;
  k.out_img = d->out_img; k.out_lab = d->out_lab; k.lab_lut = d->lab_lut;
  const int64_t total = (int64_t)d->hc * d->wc;
  hipLaunchKernelGGL(seg_sync_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream, k);
  MYOLO_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------ ColorJitter
// torchvision.transforms.ColorJitter on a PIL image (get_citys_loader, SegmentationDataset.py:462-466), restated from Pillow
// (ImageEnhance.py, Blend.c, Convert.c) and torchvision's functional_pil.py:
//   brightness f : Image.blend(black, img, f)                     out = (u8)(d + f*(i - d)) in float32, clamped when f is outside [0,1]
//   contrast   f : blend(grey(mean), img, f), mean = int(mean(L) + 0.5) of the CURRENT image, L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16
//   saturation f : blend(L replicated, img, f)
//   hue        s : RGB -> HSV (Convert.c rgb2hsv_row, float/double mix as written there), H += s (mod 256), HSV -> RGB
// applied in the caller's order.  The contrast mean needs the whole image: pass 1 applies the stages in front of it and sums L, pass 2
// applies everything and finishes with ToTensor (uint8 -> lut[v] = v/255 in the output dtype, HWC -> CHW) when asked.
namespace {

struct Jitter { int order[4]; float f[3]; int hue; };        // order: 0 brightness, 1 contrast, 2 saturation, 3 hue, -1 skip

__device__ __forceinline__ int to_l(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }
__device__ __forceinline__ int blend1(int d, int i, float f, bool inside) {
  const float t = __fadd_rn((float)d, __fmul_rn(f, (float)(i - d)));          // no FMA contraction: Blend.c is plain C on x86
  if (inside) return (int)(uint8_t)t;
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)(uint8_t)t);
}
__device__ __forceinline__ double c_round(double x) { return x >= 0.0 ? floor(x + 0.5) : ceil(x - 0.5); }

__device__ void hue_shift(int& r, int& g, int& b, int shift) {
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  int uh = 0, us = 0;
  const int uv = maxc;
  if (minc != maxc) {
    const float cr = (float)(maxc - minc);
    const float s = __fdiv_rn(cr, (float)maxc);
    const float rc = __fdiv_rn((float)(maxc - r), cr), gc = __fdiv_rn((float)(maxc - g), cr), bc = __fdiv_rn((float)(maxc - b), cr);
    float h;
    if (r == maxc) h = __fsub_rn(bc, gc);
    else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
    else h = (float)(4.0 + (double)gc - (double)rc);
    const double hd = (double)h / 6.0 + 1.0;
    h = (float)(hd - floor(hd));                                             // fmod(x, 1.0) of a positive x
    uh = clip8((int)((double)h * 255.0));
    us = clip8((int)((double)s * 255.0));
  }
  uh = (uh + shift) & 255;
  if (us == 0) { r = g = b = uv; return; }
  const double hd = (double)(float)uh * 6.0 / 255.0;
  const int i = (int)floor(hd);
  const float f = (float)(hd - (double)(float)i);
  const float fs = (float)((double)(float)us / 255.0);
  const double vf = (double)(float)uv;
  const int p = clip8((int)c_round(vf * (1.0 - (double)fs)));
  const int q = clip8((int)c_round(vf * (1.0 - (double)__fmul_rn(fs, f))));
  const int t = clip8((int)c_round(vf * __dsub_rn(1.0, __dmul_rn((double)fs, __dsub_rn(1.0, (double)f)))));   // (no FMA contraction)
  switch (i % 6) {
    case 0: r = uv; g = t; b = p; break;
    case 1: r = q; g = uv; b = p; break;
    case 2: r = p; g = uv; b = t; break;
    case 3: r = p; g = q; b = uv; break;
    case 4: r = t; g = p; b = uv; break;
    default: r = uv; g = p; b = q; break;
  }
}

// stages [0, upto) of the order; `mean`: the grey level of the contrast stage
__device__ __forceinline__ void jitter_px(const Jitter& j, int upto, int mean, int& r, int& g, int& b) {
  for (int k = 0; k < upto; ++k) {
    const int op = j.order[k];
    if (op == 0 || op == 1 || op == 2) {
      const float f = j.f[op];
      const bool inside = f >= 0.f && f <= 1.f;
      int d0, d1, d2;
      if (op == 0) d0 = d1 = d2 = 0;
      else if (op == 1) d0 = d1 = d2 = mean;
      else d0 = d1 = d2 = to_l(r, g, b);
      r = blend1(d0, r, f, inside); g = blend1(d1, g, f, inside); b = blend1(d2, b, f, inside);
    } else if (op == 3) {
      hue_shift(r, g, b, j.hue);
    }
  }
}

__global__ __launch_bounds__(256) void jitter_sum_kernel(const uint8_t* __restrict__ img, int64_t total, Jitter j, int upto,
                                                         unsigned long long* sum) {
  __shared__ unsigned long long sh[4];
  unsigned long long acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int r = img[i * 3], g = img[i * 3 + 1], b = img[i * 3 + 2];
    jitter_px(j, upto, 0, r, g, b);
    acc += (unsigned long long)to_l(r, g, b);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sum, sh[0] + sh[1] + sh[2] + sh[3]);
}

template <typename D>
__global__ __launch_bounds__(256) void jitter_apply_kernel(const uint8_t* __restrict__ img, int64_t total, Jitter j,
                                                           const unsigned long long* sum, uint8_t* out_u8, D* out_f,
                                                           const D* __restrict__ lut_g) {
  __shared__ D lut[256];
  if (out_f) lut[threadIdx.x] = lut_g[threadIdx.x];
  __syncthreads();
  // ImageStat.Stat(img.convert('L')).mean[0] = sum / count in double; ImageEnhance.Contrast: int(mean + 0.5)
  const int mean = sum ? (int)((double)sum[0] / (double)total + 0.5) : 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int r = img[i * 3], g = img[i * 3 + 1], b = img[i * 3 + 2];
    jitter_px(j, 4, mean, r, g, b);
    if (out_u8) { out_u8[i * 3] = (uint8_t)r; out_u8[i * 3 + 1] = (uint8_t)g; out_u8[i * 3 + 2] = (uint8_t)b; }
    if (out_f) { out_f[i] = lut[r]; out_f[total + i] = lut[g]; out_f[2 * total + i] = lut[b]; }
  }
}

}  // namespace

extern "C" int myolo_color_jitter(const uint8_t* img_hwc, int h, int w, const int32_t* order4, float brightness, float contrast,
                                  float saturation, int hue_shift_u8, uint64_t* scratch, uint8_t* out_hwc, void* out_chw,
                                  int out_dtype, const void* lut256, void* stream) {
  if (!img_hwc || h < 1 || w < 1 || !order4 || (!out_hwc && !out_chw)) return MYOLO_EINVAL;
  if (out_chw && (!lut256 || (out_dtype != MYOLO_F16 && out_dtype != MYOLO_F32))) return MYOLO_EINVAL;
  Jitter j;
  int cpos = -1, seen = 0;
  for (int k = 0; k < 4; ++k) {
    const int op = order4[k];
    if (op < -1 || op > 3) return MYOLO_EINVAL;
    if (op >= 0) { if (seen & (1 << op)) return MYOLO_EINVAL; seen |= 1 << op; }
    j.order[k] = op;
    if (op == 1) cpos = k;
  }
  j.f[0] = brightness; j.f[1] = contrast; j.f[2] = saturation; j.hue = hue_shift_u8 & 255;
  if (cpos >= 0 && !scratch) return MYOLO_EINVAL;
  const int64_t total = (int64_t)h * w;
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for(total, 256, 4096);
  if (cpos >= 0) {
    hipError_t e = hipMemsetAsync(scratch, 0, sizeof(uint64_t), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(jitter_sum_kernel, dim3(grid), dim3(256), 0, st, img_hwc, total, j, cpos, reinterpret_cast<unsigned long long*>(scratch));
    MYOLO_CHECK_LAUNCH();
  }
  const unsigned long long* sum = cpos >= 0 ? reinterpret_cast<const unsigned long long*>(scratch) : nullptr;
  if (out_chw && out_dtype == MYOLO_F16)
    hipLaunchKernelGGL(jitter_apply_kernel<half_t>, dim3(grid), dim3(256), 0, st, img_hwc, total, j, sum, out_hwc, (half_t*)out_chw, (const half_t*)lut256);
  else
    hipLaunchKernelGGL(jitter_apply_kernel<float>, dim3(grid), dim3(256), 0, st, img_hwc, total, j, sum, out_hwc, (float*)out_chw, (const float*)lut256);
  MYOLO_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------- detection: mosaic + warp + HSV
// utils/datasets.py:672-724 load_mosaic (4 images on a 2s x 2s grey canvas) -> :851-895 random_perspective (cv2.warpAffine, INTER_LINEAR,
// borderValue 114, output s x s) -> :646-658 augment_hsv -> :574-584 flips -> :590 BGR->RGB, HWC->CHW, fused: an output pixel is traced
// back through the affine map to (at most) four canvas pixels, each of which is a pixel of one source image or the grey border; the
// 2s x 2s canvas never exists.  OpenCV is not installed here: its 8-bit arithmetic is restated (imgwarp.cpp warpAffine + remapBilinear:
// AB_BITS 10, INTER_BITS 5, weights (32-a)(32-b)*32 summing to 2^15; color_hsv RGB2HSV_b integer tables / HSV2RGB float) -- "parity
// unpinned", as for letterbox's cv2.resize.
namespace {

struct MosaicSrc { const uint8_t* img; int h, w; int x1a, y1a, x2a, y2a; int padw, padh; };
struct MosaicWarp {
  MosaicSrc src[4]; int nsrc;
  int cw, ch;                          // canvas size
  double M[6];                         // dst -> src affine, inverted by the host the way cv::warpAffine inverts it
  int warp;                            // 0: the canvas is copied (identity matrix and no border: random_perspective's `image changed` test)
  int ow, oh;
  const uint8_t* lut;                  // [3][256] hue / sat / val tables of augment_hsv, or NULL
  int fliplr, flipud, fill;
  uint8_t* out_chw; uint8_t* out_hwc;
};

__device__ __forceinline__ void canvas_px(const MosaicWarp& p, int X, int Y, int& b, int& g, int& r) {
  b = g = r = p.fill;
  if ((unsigned)X >= (unsigned)p.cw || (unsigned)Y >= (unsigned)p.ch) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k >= p.nsrc) break;
    const MosaicSrc& s = p.src[k];
    if (X >= s.x1a && X < s.x2a && Y >= s.y1a && Y < s.y2a) {
      const uint8_t* q = s.img + ((int64_t)(Y - s.padh) * s.w + (X - s.padw)) * 3;
      b = q[0]; g = q[1]; r = q[2];
      return;
    }
  }
}

__device__ __forceinline__ int cv_round(double v) { return (int)rint(v); }          // cvRound: round half to even

__device__ void hsv_lut_bgr(int& b, int& g, int& r, const uint8_t* lut, const int* sdiv, const int* hdiv) {
  // cv2.cvtColor(BGR2HSV) on uint8 (H in [0,180))
  int v = max(b, max(g, r)), vmin = min(b, min(g, r));
  const int diff = v - vmin;
  const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
  int s = (diff * sdiv[v] + (1 << 11)) >> 12;
  int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
  h = (h * hdiv[diff] + (1 << 11)) >> 12;
  h += h < 0 ? 180 : 0;
  h = clip8(h);
  // cv2.LUT per plane
  h = lut[h]; s = lut[256 + (s & 255)]; v = lut[512 + v];
  // cv2.cvtColor(HSV2BGR) on uint8: float path
  const float fs = (float)s * (1.f / 255.f), fv = (float)v * (1.f / 255.f);
  float fb, fg, fr;
  if (s == 0) fb = fg = fr = fv;
  else {
    float fh = (float)h * (6.f / 180.f);
    fh = fmodf(fh, 6.f);
    int sector = (int)floorf(fh);
    fh -= (float)sector;
    if ((unsigned)sector >= 6u) { sector = 0; fh = 0.f; }
    float tab[4];
    tab[0] = fv; tab[1] = fv * (1.f - fs); tab[2] = fv * (1.f - fs * fh); tab[3] = fv * (1.f - fs * (1.f - fh));
    const int sd[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    fb = tab[sd[sector][0]]; fg = tab[sd[sector][1]]; fr = tab[sd[sector][2]];
  }
  b = clip8((int)rintf(fb * 255.f)); g = clip8((int)rintf(fg * 255.f)); r = clip8((int)rintf(fr * 255.f));
}

__global__ __launch_bounds__(256) void mosaic_warp_kernel(const MosaicWarp p) {
  __shared__ int sdiv[256], hdiv[256];
  {
    const int i = threadIdx.x;
    sdiv[i] = i ? cv_round((double)(255 << 12) / (1. * i)) : 0;
    hdiv[i] = i ? cv_round((double)(180 << 12) / (6. * i)) : 0;
  }
  __syncthreads();
  const int64_t total = (int64_t)p.oh * p.ow;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int y = (int)(i / p.ow), x = (int)(i - (int64_t)y * p.ow);
    int b, g, r;
    if (!p.warp) canvas_px(p, x, y, b, g, r);
    else {
      const int adx = cv_round(p.M[0] * x * 1024.), bdx = cv_round(p.M[3] * x * 1024.);
      const int X0 = cv_round((p.M[1] * y + p.M[2]) * 1024.) + 16, Y0 = cv_round((p.M[4] * y + p.M[5]) * 1024.) + 16;
      const int X = (X0 + adx) >> 5, Y = (Y0 + bdx) >> 5;
      int sx = X >> 5, sy = Y >> 5;
      sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx); sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);   // saturate_cast<short>
      const int fa = Y & 31, fb_ = X & 31;
      const int w00 = (32 - fa) * (32 - fb_) * 32, w01 = (32 - fa) * fb_ * 32, w10 = fa * (32 - fb_) * 32, w11 = fa * fb_ * 32;
      int b0, g0, r0, b1, g1, r1, b2, g2, r2, b3, g3, r3;
      canvas_px(p, sx, sy, b0, g0, r0); canvas_px(p, sx + 1, sy, b1, g1, r1);
      canvas_px(p, sx, sy + 1, b2, g2, r2); canvas_px(p, sx + 1, sy + 1, b3, g3, r3);
      b = clip8((b0 * w00 + b1 * w01 + b2 * w10 + b3 * w11 + (1 << 14)) >> 15);
      g = clip8((g0 * w00 + g1 * w01 + g2 * w10 + g3 * w11 + (1 << 14)) >> 15);
      r = clip8((r0 * w00 + r1 * w01 + r2 * w10 + r3 * w11 + (1 << 14)) >> 15);
    }
    if (p.lut) hsv_lut_bgr(b, g, r, p.lut, sdiv, hdiv);
    const int oy = p.flipud ? p.oh - 1 - y : y, ox = p.fliplr ? p.ow - 1 - x : x;
    const int64_t o = (int64_t)oy * p.ow + ox;
    if (p.out_chw) { p.out_chw[o] = (uint8_t)r; p.out_chw[total + o] = (uint8_t)g; p.out_chw[2 * total + o] = (uint8_t)b; }
    if (p.out_hwc) { p.out_hwc[o * 3] = (uint8_t)b; p.out_hwc[o * 3 + 1] = (uint8_t)g; p.out_hwc[o * 3 + 2] = (uint8_t)r; }
  }
}

// cv2.resize(img, (rw, rh), interpolation=cv2.INTER_LINEAR) on uint8 HWC (load_image, datasets.py:638-640): the arithmetic of
// frame.hip's letterbox resampler, HWC in -> HWC out
__global__ __launch_bounds__(256) void resize_u8_kernel(const uint8_t* __restrict__ im, int h0, int w0, int rh, int rw, uint8_t* __restrict__ out,
                                                        double scale_x, double scale_y, int area2) {
  const int64_t total = (int64_t)rh * rw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int dy = (int)(i / rw), dx = (int)(i - (int64_t)dy * rw);
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
      const int a0 = (int)rintf((1.f - fx) * 2048.f), a1 = (int)rintf(fx * 2048.f), b0 = (int)rintf((1.f - fy) * 2048.f), b1 = (int)rintf(fy * 2048.f);
      const int x1 = sx + 1 < w0 ? sx + 1 : w0 - 1;
      const int y0 = sy < 0 ? 0 : (sy > h0 - 1 ? h0 - 1 : sy), y1 = sy + 1 < 0 ? 0 : (sy + 1 > h0 - 1 ? h0 - 1 : sy + 1);
      const uint8_t* r0 = im + (int64_t)y0 * w0 * 3;
      const uint8_t* r1 = im + (int64_t)y1 * w0 * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int h0v = r0[sx * 3 + c] * a0 + r0[x1 * 3 + c] * a1;
        const int h1v = r1[sx * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
        c3[c] = clip8((((b0 * (h0v >> 4)) >> 16) + ((b1 * (h1v >> 4)) >> 16) + 2) >> 2);
      }
    }
    out[i * 3] = (uint8_t)c3[0]; out[i * 3 + 1] = (uint8_t)c3[1]; out[i * 3 + 2] = (uint8_t)c3[2];
  }
}

}  // namespace

extern "C" int myolo_resize_u8(const uint8_t* img_hwc, int h0, int w0, int rh, int rw, uint8_t* out_hwc, void* stream) {
  if (!img_hwc || !out_hwc || h0 < 1 || w0 < 1 || rh < 1 || rw < 1) return MYOLO_EINVAL;
  const int area2 = (w0 == 2 * rw && h0 == 2 * rh) ? 1 : 0;
  hipLaunchKernelGGL(resize_u8_kernel, dim3(grid_for((int64_t)rh * rw, 256, 4096)), dim3(256), 0, (hipStream_t)stream, img_hwc, h0, w0, rh, rw,
                     out_hwc, (double)w0 / rw, (double)h0 / rh, area2);
  MYOLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int myolo_mosaic_warp(const myolo_mosaic_desc* d, void* stream) {
  if (!d || d->nsrc < 1 || d->nsrc > 4 || d->cw < 1 || d->ch < 1 || d->ow < 1 || d->oh < 1 || (!d->out_chw && !d->out_hwc)) return MYOLO_EINVAL;
  MosaicWarp k;
  k.nsrc = d->nsrc;
  for (int i = 0; i < d->nsrc; ++i) {
    const myolo_mosaic_src& s = d->src[i];
    if (!s.img || s.h < 1 || s.w < 1) return MYOLO_EINVAL;
    // the pasted window must lie inside both the canvas and the source image
    if (s.x1a < 0 || s.y1a < 0 || s.x2a > d->cw || s.y2a > d->ch || s.x1a - s.padw < 0 || s.y1a - s.padh < 0 ||
        s.x2a - s.padw > s.w || s.y2a - s.padh > s.h)
      return MYOLO_EINVAL;
    k.src[i] = MosaicSrc{s.img, s.h, s.w, s.x1a, s.y1a, s.x2a, s.y2a, s.padw, s.padh};
  }
  k.cw = d->cw; k.ch = d->ch; k.warp = d->warp; k.ow = d->ow; k.oh = d->oh;
  for (int i = 0; i < 6; ++i) k.M[i] = d->M[i];
  if (!d->warp && (d->ow != d->cw || d->oh != d->ch)) return MYOLO_EINVAL;
  k.lut = d->hsv_lut; k.fliplr = d->fliplr; k.flipud = d->flipud; k.fill = d->fill & 255;
  k.out_chw = d->out_chw; k.out_hwc = d->out_hwc;
  hipLaunchKernelGGL(mosaic_warp_kernel, dim3(grid_for((int64_t)d->oh * d->ow, 256, 8192)), dim3(256), 0, (hipStream_t)stream, k);
  MYOLO_CHECK_LAUNCH();
  return 0;
}
