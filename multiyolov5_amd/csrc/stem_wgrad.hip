// The stem layer's BatchNorm backward AND weight gradient in one pass over (gout, y, x) -- round 6.
//
// The first Conv of the network (Focus: 3x3, 12 -> 32 channels on the 256x512 map; reference models/common.py:42-43,540-551) has no input
// gradient, so its dy = BatchNorm/SiLU-backward(gout) has exactly one reader: its own weight gradient.  Until round 6 the END of every backward
// pass -- the critical tail in front of the optimizer -- was three dependent launches over the step's largest tensors (16 x 256 x 512 x 32 =
// 134 MB each): bn_act_bwd_reduce (gout, y read: 84 us), bn_act_bwd_apply (gout, y read, dy written: 108 us), then the weight gradient (dy, x
// read: 43 + 8 us).  But dy is LINEAR in quantities that need no second pass:
//     g = gout * act'(z)            xhat = (y - mean) * invstd                                  (mean, invstd: saved by the forward)
//     dy = sc * (g - k0 - k1 * xhat)        k0 = sum(g) / M,  k1 = sum(g * xhat) / M,  sc = gamma * invstd
//     dW[co][ci][t] = sum_p dy[p][co] * x[p + tap_t][ci] = sc[co] * ( A[t][co][ci] - k0[co] * S[t][ci] - k1[co] * B[t][co][ci] )
//     A = sum_p g (x) x,   B = sum_p xhat (x) x,   S[t][ci] = sum_p x[p + tap_t][ci]
// so ONE pass that reads gout, y and x once produces A, B, S and the two BatchNorm sums; a 1-workgroup epilogue combines them into dW, dgamma,
// dbeta.  dy never exists: 268 MB of reads and 134 MB of writes per step disappear together with two launches of the tail.
//
// Kernel (the tile scheme of conv_wgrad.hip's wgrad_small_halo_kernel): a workgroup walks 8 x 16-pixel tiles; per tile every thread loads its
// 16-byte pieces of gout and y (tile pixel, 8 channels) and of the 10 x 18 x halo, turns (gout, y) into (g, xhat) in registers -- adding to
// its per-channel partial sums of g and g * xhat -- and stages [g | xhat] (64 fp16 rows of the MFMA's A operand) and the halo in LDS; wave w
// owns row fragment w (g 0-15, g 16-31, xhat 0-15, xhat 16-31) for all 128 pixels x 9 taps (36 MFMAs per tile and wave, fragments by
// ds_read_b64_tr_b16), so the waves' accumulators cover disjoint outputs and need no cross-wave reduction.  S costs nothing on the matrix
// pipe: a loader thread's halo position is the same in every tile, so it keeps ONE running sum of its 8 x channels; the epilogue knows
// which taps a halo position feeds.  Split-K over tiles (768 workgroups, three per CU), partials to a workspace, one reduce launch, one
// 1-workgroup combine.
#include "myolo_dev.h"
#include <string.h>
#include <stdlib.h>

namespace stem {

constexpr int THREADS = 256, TH = 8, TW = 16, HH = TH + 2, HW = TW + 2, NT = 9, KPS = TH * TW;
constexpr int PD = 64 * 2 + 16;                 // staged A rows: g[32] | xhat[32] halves + pad (odd number of 16-byte units: conflict-free transpose reads)
constexpr int PX = 16 * 2 + 16;                 // staged x halo rows: 16 halves + pad
constexpr int NSLOT = HH * HW * 2;              // x loader slots: (halo pixel, 8-channel segment)
constexpr int WS_ACC = 4 * NT * 256;            // [row fragment][tap][lane][4] fp32
constexpr int WS_SUM = 64;                      // sum g [32], sum g * xhat [32]
constexpr int WS_SX = NSLOT * 8;                // per-slot x sums
constexpr int WS_SPLIT = WS_ACC + WS_SUM + WS_SX;

struct StemK {
  const char* x; int x_sn, x_sh, x_sw, Hi, Wi, Cin;          // element strides
  const char* g; int g_sn, g_sh, g_sw;
  const char* y; int y_sn, y_sh, y_sw;
  int Cout, N, tx, ty, ntiles, tps, ksplit;
  int x_bytes, g_bytes, y_bytes;
  const float* saved; const float* gamma; const float* beta;
  float* ws;
  int tap_dy[NT], tap_dx[NT];
};

__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void stem_bn_wgrad_kernel(const StemK p) {
  constexpr int ES = 2, SEG = 8;
  constexpr int SD_BYTES = KPS * PD, SX_BYTES = HH * HW * PX;
  static_assert(SD_BYTES + SX_BYTES >= THREADS * 16 * 4, "the sums' reduction reuses the staging area");
  __shared__ __attribute__((aligned(16))) char smem[SD_BYTES + SX_BYTES];
  char* sD = smem;
  char* sX = smem + SD_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int split = blockIdx.x;
  const int t_begin = split * p.tps;
  int t_end = t_begin + p.tps;
  if (t_end > p.ntiles) t_end = p.ntiles;
  const int nsteps = t_end > t_begin ? t_end - t_begin : 0;
  const int tiles_img = p.tx * p.ty;

  f4_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f4_t{0.f, 0.f, 0.f, 0.f};

  constexpr int OOB = 0x7fff0000;
  const __amdgpu_buffer_rsrc_t rbx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rbg = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.g), 0, p.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rby = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.y), 0, p.y_bytes, 0x00020000);
  auto bl = [&](const __amdgpu_buffer_rsrc_t& r, int off) -> uint4 {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return uint4{v.x, v.y, v.z, v.w};
  };
  // loader constants: (gout, y) pieces l = 0, 1: tile pixel v >> 2, channel segment v & 3 (the SAME segment for both: 256 % 4 == 0);
  // halo pieces: halo pixel v >> 1, segment v & 1
  const int cd = (tid & 3) * SEG;
  const bool c_on = cd < p.Cout;
  int g_off[2], y_off[2], x_hy[2], x_hx[2];
  bool x_on[2];
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const int v = tid + l * THREADS;
    const int prow = v >> 2;
    g_off[l] = ((prow >> 4) * p.g_sh + (prow & 15) * p.g_sw + cd) * ES;
    y_off[l] = ((prow >> 4) * p.y_sh + (prow & 15) * p.y_sw + cd) * ES;
    const int hp = v >> 1;
    x_on[l] = hp < HH * HW && (v & 1) * SEG < p.Cin;
    x_hy[l] = hp / HW - 1; x_hx[l] = hp % HW - 1;
  }
  // per-channel constants in LDS ([4][32]: z = y * sc + sh, xhat = y * is + mb), read once per tile: 32 registers less per thread is what
  // lets three workgroups share a CU without scratch (the kernel is latency-bound: 2 per CU measured 160 us, HBM needs 85)
  __shared__ float sK[4][32];
  if (tid < 32) {
    const bool in = tid < p.Cout;
    const float mean = in ? p.saved[tid] : 0.f, istd = in ? p.saved[p.Cout + tid] : 0.f;
    const float scv = in ? p.gamma[tid] * istd : 0.f;
    sK[0][tid] = scv; sK[1][tid] = in ? p.beta[tid] - mean * scv : 0.f; sK[2][tid] = istd; sK[3][tid] = -mean * istd;
  }
  __syncthreads();
  float s0[SEG], s1[SEG], sx[2][SEG];
#pragma unroll
  for (int i = 0; i < SEG; ++i) { s0[i] = 0.f; s1[i] = 0.f; sx[0][i] = 0.f; sx[1][i] = 0.f; }
  // two tiles of loads in flight per thread (as wgrad_small_halo_kernel: the MFMAs of a tile take a fraction of the HBM latency)
  uint4 rg[2][2], ry[2][2], rx[2][2];
  auto issue = [&](int s, const int b) {
    const int tl = t_begin + s;
    const int n = tl / tiles_img; const int r = tl - n * tiles_img; const int by = r / p.tx; const int bx = r - by * p.tx;
    const int oy0 = by * TH, ox0 = bx * TW;
    const int gbase = (n * p.g_sn + oy0 * p.g_sh + ox0 * p.g_sw) * ES;
    const int ybase = (n * p.y_sn + oy0 * p.y_sh + ox0 * p.y_sw) * ES;
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      rg[b][l] = bl(rbg, c_on ? gbase + g_off[l] : OOB);
      ry[b][l] = bl(rby, c_on ? ybase + y_off[l] : OOB);
    }
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const int iy = oy0 + x_hy[l], ix = ox0 + x_hx[l];
      const bool ok = x_on[l] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      rx[b][l] = bl(rbx, ok ? (n * p.x_sn + iy * p.x_sh + ix * p.x_sw + ((tid + l * THREADS) & 1) * SEG) * ES : OOB);
    }
  };
  // fragment addresses (conv_wgrad.hip): K index 8g + 4h + k' of a 32-pixel K step <-> tile pixel ks*32 + 8g + 4h + k'
  const int g = lane >> 4, kq = (lane & 15) >> 2, q = lane & 3;
  const int ca = (wave * 16 + q * 4) * 2;                  // this wave's 16 rows of [g | xhat]
  int tap_b[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) tap_b[t] = (p.tap_dy[t] * HW + p.tap_dx[t]) * PX;

  auto step = [&](int s, const int b) {
    float sc[SEG], sh[SEG], is[SEG], mb[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) { sc[i] = sK[0][cd + i]; sh[i] = sK[1][cd + i]; is[i] = sK[2][cd + i]; mb[i] = sK[3][cd + i]; }
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const int v = tid + l * THREADS;
      // (one piece at a time: the packed registers become "new" values behind the previous piece's last accumulate, or hipcc converts and
      //  transforms both pieces side by side -- 32 more live registers)
      asm volatile("" : "+v"(rg[b][l].x), "+v"(rg[b][l].y), "+v"(rg[b][l].z), "+v"(rg[b][l].w), "+v"(s0[0]), "+v"(s1[SEG - 1]));
      asm volatile("" : "+v"(ry[b][l].x), "+v"(ry[b][l].y), "+v"(ry[b][l].z), "+v"(ry[b][l].w), "+v"(s0[0]));
      float fg[SEG], fy[SEG], gv[SEG], xh[SEG];
      Vec<half_t>::unpack(rg[b][l], fg); Vec<half_t>::unpack(ry[b][l], fy);
#pragma unroll
      for (int i = 0; i < SEG; ++i) {
        gv[i] = fg[i] * silu_grad_f(fmaf(fy[i], sc[i], sh[i]));
        xh[i] = fmaf(fy[i], is[i], mb[i]);
        s0[i] += gv[i];
        s1[i] += gv[i] * xh[i];
      }
      const u32x4_t pg = pack_h8(gv), ph = pack_h8(xh);
      *reinterpret_cast<uint4*>(&sD[(v >> 2) * PD + cd * 2]) = uint4{pg.x, pg.y, pg.z, pg.w};
      *reinterpret_cast<uint4*>(&sD[(v >> 2) * PD + 64 + cd * 2]) = uint4{ph.x, ph.y, ph.z, ph.w};
      if ((v >> 1) < HH * HW) {
        float fx[SEG];
        Vec<half_t>::unpack(rx[b][l], fx);
#pragma unroll
        for (int i = 0; i < SEG; ++i) sx[l][i] += fx[i];
        *reinterpret_cast<uint4*>(&sX[(v >> 1) * PX + (v & 1) * 16]) = rx[b][l];
      }
    }
    __syncthreads();
    if (s + 2 < nsteps) issue(s + 2, b);                 // (the registers just consumed are free)
#pragma unroll 1                                          // (unrolled, hipcc hoists all 72 fragment reads of a tile: 256 VGPRs + scratch)
    for (int ks = 0; ks < 4; ++ks) {
      const int pr = ks * 32 + 8 * g + kq;
      const int hb = ((pr >> 4) + 1) * HW + (pr & 15) + 1;   // its halo pixel at tap (0, 0)
      h8_t fa;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        fp16x4_t va = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(&sD[(pr + 4 * h) * PD + ca]));
#pragma unroll
        for (int e = 0; e < 4; ++e) fa[4 * h + e] = (half_t)va[e];
      }
      // all nine taps' fragments in flight before the first MFMA (one tap at a time was a read -> ~300-cycle wait -> one MFMA chain:
      // 36 of them per tile and wave)
      h8_t fb[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          fp16x4_t vb = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
              (__attribute__((address_space(3))) fp16x4_t*)(&sX[(hb + 4 * h) * PX + tap_b[t] + q * 8]));
#pragma unroll
          for (int e = 0; e < 4; ++e) fb[t][4 * h + e] = (half_t)vb[e];
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb[t], acc[t], 0, 0, 0);
    }
    __syncthreads();
  };
  if (nsteps > 0) issue(0, 0);
  if (nsteps > 1) issue(1, 1);
  for (int s = 0; s < nsteps; s += 2) {
    step(s, 0);
    if (s + 1 < nsteps) step(s + 1, 1);
  }
  // ---- partials of this split: accumulators lane-linear (64-byte-coalesced f4 stores), then the sums
  float* dst = p.ws + (int64_t)split * WS_SPLIT;
#pragma unroll
  for (int t = 0; t < NT; ++t) reinterpret_cast<f4_t*>(dst)[(wave * NT + t) * 64 + lane] = acc[t];
  float* red = reinterpret_cast<float*>(smem);             // [THREADS][16]
#pragma unroll
  for (int i = 0; i < SEG; ++i) { red[tid * 16 + i] = s0[i]; red[tid * 16 + 8 + i] = s1[i]; }
  __syncthreads();
  if (tid < 64) {                                          // tid = which * 32 + channel
    const int which = tid >> 5, c = tid & 31, seg = c >> 3, i = c & 7;
    float a = 0.f;
    for (int k = 0; k < THREADS / 4; ++k) a += red[(seg + 4 * k) * 16 + which * 8 + i];
    dst[WS_ACC + tid] = a;
  }
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const int v = tid + l * THREADS;
    if (v < NSLOT) {
      float* o = dst + WS_ACC + WS_SUM + v * 8;
      *reinterpret_cast<f4_t*>(o) = f4_t{sx[l][0], sx[l][1], sx[l][2], sx[l][3]};
      *reinterpret_cast<f4_t*>(o + 4) = f4_t{sx[l][4], sx[l][5], sx[l][6], sx[l][7]};
    }
  }
}

// totals[e] = sum over the splits of ws[s][e]: a workgroup owns 64 consecutive entries, its 4 waves take every 4th split, 8 loads in flight per
// lane (two in flight: 23 us for 512 splits -- one L2 round trip per pair)
__global__ __launch_bounds__(256) void stem_reduce_kernel(const float* __restrict__ ws, float* __restrict__ tot, int ks) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (e < WS_SPLIT) {
    int s = wave;
    for (; s + 28 < ks; s += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += ws[(int64_t)(s + 4 * u) * WS_SPLIT + e];
    }
    for (; s < ks; s += 4) a[0] += ws[(int64_t)s * WS_SPLIT + e];
  }
  red[wave][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (wave == 0 && e < WS_SPLIT) tot[e] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// dW[co][ci][t] += sc[co] * (A - k0[co] * S[t][ci] - k1[co] * B); dgamma += sum g * xhat; dbeta += sum g.  One workgroup; the totals (48 KB)
// go through LDS once (reading them where needed was 180 dependent L2 loads per thread: 32 us).
__global__ __launch_bounds__(256) void stem_combine_kernel(const float* __restrict__ tot, const StemK p, float* dw, float* dgamma, float* dbeta, int cout_w,
                                                           int cin_w, float rM) {
  __shared__ __attribute__((aligned(16))) float T[WS_SPLIT];
  __shared__ float S[NT][16];
  __shared__ float k0[32], k1[32], scs[32];
  const int tid = threadIdx.x;
  static_assert(WS_SPLIT % 4 == 0, "vector copy");
  for (int j = tid; j < WS_SPLIT / 4; j += 256) reinterpret_cast<f4_t*>(T)[j] = reinterpret_cast<const f4_t*>(tot)[j];
  __syncthreads();
  for (int j = tid; j < NT * 16; j += 256) {
    const int t = j >> 4, ci = j & 15;
    float a = 0.f;
    for (int hp = 0; hp < HH * HW; ++hp) {                 // halo pixel (hy, hx) feeds tap t iff the output pixel (hy - dy, hx - dx) is inside the tile
      const int oy = hp / HW - 1 - p.tap_dy[t], ox = hp % HW - 1 - p.tap_dx[t];
      if (oy >= 0 && oy < TH && ox >= 0 && ox < TW) a += T[WS_ACC + WS_SUM + (hp * 2 + (ci >> 3)) * 8 + (ci & 7)];
    }
    S[t][ci] = a;
  }
  if (tid < 32) {
    const bool in = tid < p.Cout;
    const float d0 = in ? T[WS_ACC + tid] : 0.f, d1 = in ? T[WS_ACC + 32 + tid] : 0.f;
    k0[tid] = d0 * rM; k1[tid] = d1 * rM;
    scs[tid] = in ? p.gamma[tid] * p.saved[p.Cout + tid] : 0.f;
    if (in) {
      if (dgamma) dgamma[tid] += d1;
      if (dbeta) dbeta[tid] += d0;
    }
  }
  __syncthreads();
  for (int e = tid; e < NT * cout_w * cin_w; e += 256) {
    const int t = e % NT, ci = (e / NT) % cin_w, co = e / (NT * cin_w);
    // accumulator image: [row fragment f][tap][lane][r] with row = 4 * (lane >> 4) + r, column ci = lane & 15; g rows: f = co >> 4, xhat rows: 2 + (co >> 4)
    const int row = co & 15, ln = (row >> 2) * 16 + ci, r = row & 3;
    const float A = T[(((co >> 4) * NT + t) * 64 + ln) * 4 + r];
    const float B = T[(((2 + (co >> 4)) * NT + t) * 64 + ln) * 4 + r];
    dw[e] += scs[co] * (A - k0[co] * S[t][ci] - k1[co] * B);
  }
}

}  // namespace stem

static int g_stem = -1;                  // MYOLO_STEM_WGRAD=0: never (the three launches)
static int g_stem_ks = 512;              // workgroups of the main kernel: two per CU (243 VGPRs)
int myolo_stem_set(const char* name, int value) {
  if (!strcmp(name, "stem_wgrad")) { g_stem = value; return 0; }
  if (!strcmp(name, "stem_ks")) { g_stem_ks = value; return 0; }
  return MYOLO_EINVAL;
}

static bool stem_ok(const myolo_wgrad_desc* d, const myolo_tensor* gout, const myolo_tensor* y) {
  using namespace stem;
  if (g_stem < 0) g_stem = getenv("MYOLO_STEM_WGRAD") ? atoi(getenv("MYOLO_STEM_WGRAD")) : 1;
  if (!g_stem || !d || !gout || !d->x.ptr || !gout->ptr || !d->dw || d->db) return false;
  if (d->x.dtype != MYOLO_F16 || gout->dtype != MYOLO_F16 || d->ntaps != NT || d->stride != 1 || d->up_shift != 0) return false;
  const int cout = d->cout > 0 ? d->cout : gout->c, cin = d->cin > 0 ? d->cin : d->x.c;
  if (cout > 32 || cout % 8 || gout->c != cout || cin > 16 || d->x.c > 16 || (d->x.c > 8 && d->x.sw < 16)) return false;
  if (d->x.n != gout->n || d->x.h != gout->h || d->x.w != gout->w || gout->h % TH || gout->w % TW) return false;
  bool seen[9] = {false};
  for (int t = 0; t < NT; ++t) {
    const int dy = d->tap_dy[t], dx = d->tap_dx[t];
    if (dy < -1 || dy > 1 || dx < -1 || dx > 1 || seen[(dy + 1) * 3 + dx + 1]) return false;
    seen[(dy + 1) * 3 + dx + 1] = true;
  }
  auto fits = [](const myolo_tensor& t) {
    return ((int64_t)t.n * t.sn + (int64_t)t.h * t.sh + (int64_t)t.w * t.sw + t.c) * 2 < (1ll << 31) && !((uintptr_t)t.ptr & 15) && !(t.sw % 8) && !(t.sh % 8) && !(t.sn % 8);
  };
  if (!fits(d->x) || !fits(*gout)) return false;
  if (y && (!y->ptr || y->dtype != MYOLO_F16 || y->n != gout->n || y->h != gout->h || y->w != gout->w || y->c != gout->c || !fits(*y))) return false;
  return true;
}

extern "C" int myolo_bn_wgrad_stem_ok(const myolo_wgrad_desc* d, const myolo_tensor* gout) { return stem_ok(d, gout, nullptr) ? 1 : 0; }
extern "C" int64_t myolo_bn_wgrad_stem_ws_bytes(void) { return (int64_t)(1024 + 1) * stem::WS_SPLIT * (int64_t)sizeof(float); }

extern "C" int myolo_bn_wgrad_stem(const myolo_wgrad_desc* d, const myolo_tensor* gout, const myolo_tensor* y, const float* saved, const float* gamma,
                                   const float* beta, int act, float* dgamma, float* dbeta, float* ws, int64_t ws_bytes, void* stream) {
  using namespace stem;
  if (!saved || !gamma || !beta || !ws || ((uintptr_t)ws & 15) || act != MYOLO_ACT_SILU || !stem_ok(d, gout, y)) return MYOLO_EINVAL;
  StemK k;
  k.x = (const char*)d->x.ptr; k.x_sn = (int)d->x.sn; k.x_sh = (int)d->x.sh; k.x_sw = (int)d->x.sw; k.Hi = d->x.h; k.Wi = d->x.w; k.Cin = d->x.c;
  k.g = (const char*)gout->ptr; k.g_sn = (int)gout->sn; k.g_sh = (int)gout->sh; k.g_sw = (int)gout->sw;
  k.y = (const char*)y->ptr; k.y_sn = (int)y->sn; k.y_sh = (int)y->sh; k.y_sw = (int)y->sw;
  k.Cout = gout->c; k.N = gout->n; k.tx = gout->w / TW; k.ty = gout->h / TH;
  k.ntiles = k.N * k.tx * k.ty;
  auto bytes = [](const myolo_tensor& t) { return (int)((((int64_t)t.n - 1) * t.sn + ((int64_t)t.h - 1) * t.sh + ((int64_t)t.w - 1) * t.sw + t.c) * 2); };
  k.x_bytes = (int)((((int64_t)d->x.n - 1) * d->x.sn + ((int64_t)d->x.h - 1) * d->x.sh + ((int64_t)d->x.w - 1) * d->x.sw + (d->x.c > 8 ? 16 : 8)) * 2);
  k.g_bytes = bytes(*gout); k.y_bytes = bytes(*y);
  k.saved = saved; k.gamma = gamma; k.beta = beta;
  int ks = d->ksplit > 0 ? d->ksplit : g_stem_ks;
  if (ks > 1024) ks = 1024;
  const int64_t fit = ws_bytes / ((int64_t)WS_SPLIT * (int64_t)sizeof(float)) - 1;      // (one more slice for the totals)
  if (ks > fit) ks = (int)fit;
  if (ks > k.ntiles) ks = k.ntiles;
  if (ks < 1) return MYOLO_EINVAL;
  k.tps = (k.ntiles + ks - 1) / ks;
  ks = (k.ntiles + k.tps - 1) / k.tps;
  k.ksplit = ks;
  k.ws = ws;
  float* tot = ws + (int64_t)ks * WS_SPLIT;
  for (int t = 0; t < NT; ++t) { k.tap_dy[t] = d->tap_dy[t]; k.tap_dx[t] = d->tap_dx[t]; }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(stem_bn_wgrad_kernel, dim3(ks), dim3(THREADS), 0, st, k);
  hipLaunchKernelGGL(stem_reduce_kernel, dim3((WS_SPLIT + 63) / 64), dim3(256), 0, st, (const float*)ws, tot, ks);
  const int cout_w = d->cout > 0 ? d->cout : gout->c, cin_w = d->cin > 0 ? d->cin : d->x.c;
  const float rM = 1.0f / ((float)gout->n * (float)gout->h * (float)gout->w);
  hipLaunchKernelGGL(stem_combine_kernel, dim3(1), dim3(256), 0, st, (const float*)tot, k, d->dw, dgamma, dbeta, cout_w, cin_w, rM);
  MYOLO_CHECK_LAUNCH();
  return 0;
}
