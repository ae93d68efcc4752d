// Split-K implicit-GEMM convolution for SMALL maps (fp16): few output pixels, long K = taps*Cin.
//
//   detect.py runs the network at batch 1 (detect.py:144): the 64x128 / 32x64 / 16x32 maps of a 1024x512 frame hold 8192 / 2048 /
//   512 pixels.  Tiled over M x N that is 8-64 workgroups for a whole layer, each walking its K loop step by step behind a
//   workgroup barrier -- 10-28 us per launch, MFMA idle, 3/4 of the CUs empty (r3a inference trace: 35 launches of
//   conv_igemm<128,64,2> at 15-18 us).  Here
//     * a workgroup owns a SMALL output tile (16*MF pixels x 16*NF channels: 256-1024 workgroups per layer) and its waves split K:
//       wave w takes K steps w, w+W, ... of the (tap, 32-channel chunk) sequence, so a 3x3 128->128 layer is 9 steps per wave
//       instead of 36 per workgroup;
//     * both MFMA operands come STRAIGHT from global memory / L2 into registers (lane = row lane&15, 16-byte K segment lane>>4: 64-byte
//       pieces of NHWC pixels and of packed weight rows) through a 3-deep register ring: no LDS staging, no barrier in the K loop,
//       waves only wait for their own loads;
//     * the partial accumulators meet once in LDS (fixed summation order: bit-reproducible), the fragment's owner wave applies the
//       epilogue (BatchNorm affine, SiLU / sigmoid, residual) and stores 8-byte NHWC pieces from registers (the MFMA is issued as
//       D^T = W . X^T: a lane holds 4 consecutive channels of one pixel).
//   Weights are re-read by every M tile -- from L2: at these sizes the whole layer (<= 2.4 MB of weights) is L2 resident.
//
// Same contract as myolo_conv (include/myolo.h); selected by myolo_conv for eval-style launches (no batch statistics, no
// accumulation) on small maps.  Replaces nn.Conv2d + folded BatchNorm2d + SiLU (models/common.py:34-46, fuseforward 45-46).
#include "myolo_dev.h"
#include <stdlib.h>
#include <string.h>

namespace small {

constexpr int RING = 3;
constexpr int OOB = 0x7fff0000;

struct ConvM {
  const char* x; int64_t x_sn, x_sh, x_sw; int Hi, Wi, Cin;
  char* y; int64_t y_sn, y_sh, y_sw; int Ho, Wo, Cout, N;
  const char* w; int cin_pad, cout_pad, wtaps, ntaps, stride, up;
  int tap_dy[MYOLO_MAX_TAPS], tap_dx[MYOLO_MAX_TAPS], tap_w[MYOLO_MAX_TAPS];
  const float* scale; const float* shift; int act;
  const char* res; int64_t r_sn, r_sh, r_sw;
  int M; int x_bytes, y_bytes, r_bytes;
};

template <int MF, int NF, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void conv_small_kernel(const ConvM p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f4_t* red = reinterpret_cast<f4_t*>(smem);                    // [WAVES][MF*NF][64]
  int* sTap = reinterpret_cast<int*>(red + WAVES * MF * NF * 64); // [3][MAX_TAPS] tap_dy, tap_dx, tap_w (a dynamically indexed kernarg
  const int tid = threadIdx.x, lane = tid & 63;                   //  array would be copied to scratch)
  for (int t = tid; t < p.ntaps; t += WAVES * 64) {
    sTap[t] = p.tap_dy[t]; sTap[MYOLO_MAX_TAPS + t] = p.tap_dx[t]; sTap[2 * MYOLO_MAX_TAPS + t] = p.tap_w[t];
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int m0 = blockIdx.x * (16 * MF), n0 = blockIdx.y * (16 * NF);
  const int kchunks = p.cin_pad / 32;
  const int nk = p.ntaps * kchunks;
  const int HWo = p.Ho * p.Wo;
  const int Hlog = p.Hi << p.up, Wlog = p.Wi << p.up;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.res ? p.res : p.y), 0, p.res ? p.r_bytes : 0, 0x00020000);

  // this lane's pixel rows (the same pixels it stores in the epilogue)
  int a_off[MF], a_y0[MF], a_x0[MF], y_off[MF], r_off[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const int m = m0 + mf * 16 + l15;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    const int n = mm / HWo; const int rem = mm - n * HWo; const int oy = rem / p.Wo; const int ox = rem - oy * p.Wo;
    a_off[mf] = ok ? n * (int)p.x_sn * 2 + lq * 16 : OOB;
    a_y0[mf] = oy * p.stride; a_x0[mf] = ox * p.stride;
    y_off[mf] = ok ? (n * (int)p.y_sn + oy * (int)p.y_sh + ox * (int)p.y_sw) * 2 : OOB;
    r_off[mf] = (ok && p.res) ? (n * (int)p.r_sn + oy * (int)p.r_sh + ox * (int)p.r_sw) * 2 : OOB;
  }
  // this lane's weight rows: packed [cout_pad][wtaps][cin_pad], rows up to cout_pad exist (zero padded)
  const char* w_row[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    int row = n0 + nf * 16 + l15;
    if (row >= p.cout_pad) row = p.cout_pad - 1;                  // (tile past the last channel: any valid row, the results are dropped)
    w_row[nf] = p.w + ((int64_t)row * p.wtaps * p.cin_pad + lq * 8) * 2;
  }

  f4_t acc[MF][NF];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

  uint4 ra[RING][MF], rb[RING][NF];
  auto issue = [&](int s, uint4* da, uint4* db) {
    // every load is issued unconditionally: finished waves / out-of-image taps / padded channels use an out-of-range buffer offset
    // (the hardware returns zeros), the weight loads of a dead step re-read step 0
    const bool live = s < nk;
    const int ss = live ? s : 0;
    const int tap = ss / kchunks, kc = ss - tap * kchunks;       // wave-uniform: scalar unit
    const int dy = sTap[tap], dx = sTap[MYOLO_MAX_TAPS + tap], tw = sTap[2 * MYOLO_MAX_TAPS + tap];
    const int c0 = kc * 32;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      int iy = a_y0[mf] + dy, ix = a_x0[mf] + dx;
      const bool ok = live && (unsigned)iy < (unsigned)Hlog && (unsigned)ix < (unsigned)Wlog && c0 + lq * 8 < p.Cin;
      iy >>= p.up; ix >>= p.up;
      const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? a_off[mf] + (iy * (int)p.x_sh + ix * (int)p.x_sw + c0) * 2 : OOB, 0, 0);
      da[mf] = uint4{v.x, v.y, v.z, v.w};
    }
    const int64_t wo = ((int64_t)tw * p.cin_pad + c0) * 2;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) db[nf] = ldg16(w_row[nf] + wo);
  };

  // software pipeline over this wave's steps wave, wave+WAVES, ...: RING-1 steps in flight
#pragma unroll
  for (int j = 0; j < RING - 1; ++j) issue(wave + j * WAVES, ra[j], rb[j]);
  for (int s0 = wave; s0 < nk; s0 += RING * WAVES) {
#pragma unroll
    for (int j = 0; j < RING; ++j) {
      const int s = s0 + j * WAVES;
      issue(s + (RING - 1) * WAVES, ra[(j + RING - 1) % RING], rb[(j + RING - 1) % RING]);
      if (s < nk) {                                                // wave-uniform
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
          for (int mf = 0; mf < MF; ++mf)      // weights as the A operand, pixels as the B operand: D[cout][pixel]
            acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const h8_t*>(&rb[j][nf]),
                                                                 *reinterpret_cast<const h8_t*>(&ra[j][mf]), acc[mf][nf], 0, 0, 0);
      }
    }
  }

  // ---- the waves' partial sums meet in LDS; fragment q is finished by wave q % WAVES ----
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) red[(wave * (MF * NF) + mf * NF + nf) * 64 + lane] = acc[mf][nf];
  __syncthreads();
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int q = mf * NF + nf;
      if ((q % WAVES) != wave) continue;                           // wave-uniform
      f4_t v = red[q * 64 + lane];
#pragma unroll
      for (int w = 1; w < WAVES; ++w) {
        const f4_t t = red[(w * (MF * NF) + q) * 64 + lane];
        v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
      }
      const int c0 = n0 + nf * 16 + 4 * lq;                        // Cout % 4 == 0 (host): whole 8-byte groups only
      const bool ok = c0 < p.Cout;
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sc = (p.scale && ok) ? p.scale[c0 + r] : 1.0f;
        const float sh = (p.shift && ok) ? p.shift[c0 + r] : 0.0f;
        o[r] = act_f(v[r] * sc + sh, p.act);
      }
      if (p.res) {
        const u32x2_t g = __builtin_amdgcn_raw_buffer_load_b64(rr, ok ? r_off[mf] + c0 * 2 : OOB, 0, 0);
        const h4_t gh = *reinterpret_cast<const h4_t*>(&g);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] += (float)gh[r];
      }
      h4_t oh;
#pragma unroll
      for (int r = 0; r < 4; ++r) oh[r] = (half_t)o[r];
      __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2_t*>(&oh), ry, ok ? y_off[mf] + c0 * 2 : OOB, 0, 0);
    }
}

template <int MF, int NF, int WAVES>
int launch(const ConvM& k, hipStream_t st) {
  const int smem = WAVES * MF * NF * 64 * 16 + 3 * MYOLO_MAX_TAPS * 4;
  auto kern = conv_small_kernel<MF, NF, WAVES>;
  MYOLO_ENSURE_DYN_SMEM(kern, smem);
  const int gx = (k.M + 16 * MF - 1) / (16 * MF), gy = (k.cout_pad + 16 * NF - 1) / (16 * NF);
  hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(WAVES * 64), smem, st, k);
  MYOLO_CHECK_LAUNCH();
  return 0;
}

}  // namespace small

static int g_small_off = -1;          // MYOLO_NO_SMALL / option "small_off"
static int g_small_force = 0;         // tests: 1..4 = force tile (4,4) (4,2) (2,2) (1,2) on every qualifying launch
static int g_small_max_tiles = -1;    // layers with more 64x64 output tiles than this (default 256: M*N <= 1 M outputs) keep the tiled / streaming
                                      // kernels: every workgroup re-reads its 64 x K operand panels from L2 (r3b trace: 32768-pixel maps ran 22 us here, 18 us tiled)
static int g_small_raw = -1;          // 1: also launches with the raw epilogue (training dgrads without accumulation); default: eval epilogues only

int myolo_conv_small_set(const char* name, int value) {
  if (!strcmp(name, "small_off")) { g_small_off = value; return 0; }
  if (!strcmp(name, "small_force")) { g_small_force = value; return 0; }
  if (!strcmp(name, "small_max_tiles")) { g_small_max_tiles = value; return 0; }
  if (!strcmp(name, "small_raw")) { g_small_raw = value; return 0; }
  return MYOLO_EINVAL;
}

// returns -1 when the layer does not qualify, else a hipError_t / 0
int myolo_conv_small_try(const myolo_conv_desc* d, void* stream) {
  using namespace small;
  if (g_small_off < 0) g_small_off = 0;                   // (myolo_set_option("small_off" / "small_max_tiles" / "small_raw"): tests and sweeps)
  if (g_small_max_tiles < 0) g_small_max_tiles = 128;   // (round 4: 256 -> 128, conv_mid takes the layers in between: 912 -> 955 FPS at 2048x1024, 1325 -> 1406 at 1024x512)
  if (g_small_raw < 0) g_small_raw = 0;
  if (g_small_off) return -1;
  if (!g_small_raw && !g_small_force && !d->scale && !d->shift && d->act == MYOLO_ACT_NONE) return -1;
  if (d->x.dtype != MYOLO_F16 || d->det_no > 0 || d->stats || d->accumulate || (d->bnb && d->nbnb > 0) || (d->y.c & 3) || d->cin_pad % 32)
    return -1;
  const int64_t M = (int64_t)d->y.n * d->y.h * d->y.w;
  if (M <= 0 || M > 0x7fffffff) return MYOLO_EINVAL;
  const int64_t t64 = ((M + 63) / 64) * ((d->cout_pad + 63) / 64);
  if (!g_small_force && t64 > g_small_max_tiles) return -1;
  auto span = [](const myolo_tensor& t) -> int64_t {
    return (((int64_t)t.n - 1) * t.sn + ((int64_t)t.h - 1) * t.sh + ((int64_t)t.w - 1) * t.sw + t.c) * 2;
  };
  const int64_t xb = span(d->x), yb = span(d->y), rb = d->res.ptr ? span(d->res) : 0;
  if (xb >= 0x7ffe0000LL || yb >= 0x7ffe0000LL || rb >= 0x7ffe0000LL) return -1;      // 32-bit buffer offsets
  ConvM k;
  k.x = (const char*)d->x.ptr; k.x_sn = d->x.sn; k.x_sh = d->x.sh; k.x_sw = d->x.sw;
  k.Hi = d->x.h; k.Wi = d->x.w; k.Cin = d->x.c;
  k.y = (char*)d->y.ptr; k.y_sn = d->y.sn; k.y_sh = d->y.sh; k.y_sw = d->y.sw;
  k.Ho = d->y.h; k.Wo = d->y.w; k.Cout = d->y.c; k.N = d->y.n;
  k.w = (const char*)d->w; k.cin_pad = d->cin_pad; k.cout_pad = d->cout_pad; k.wtaps = d->wtaps;
  k.ntaps = d->ntaps; k.stride = d->stride; k.up = d->up_shift;
  for (int i = 0; i < MYOLO_MAX_TAPS; ++i) { k.tap_dy[i] = d->tap_dy[i]; k.tap_dx[i] = d->tap_dx[i]; k.tap_w[i] = d->tap_w[i]; }
  k.scale = d->scale; k.shift = d->shift; k.act = d->act;
  k.res = (const char*)d->res.ptr; k.r_sn = d->res.sn; k.r_sh = d->res.sh; k.r_sw = d->res.sw;
  k.M = (int)M; k.x_bytes = (int)xb; k.y_bytes = (int)yb; k.r_bytes = (int)rb;
  const int nk = d->ntaps * (d->cin_pad / 32);
  // tile: the largest one that still gives every CU a workgroup (256 CUs); long K loops on the smallest tiles get 8 waves
  const int64_t mt64 = (M + 63) / 64, mt32 = (M + 31) / 32, mt16 = (M + 15) / 16;
  const int64_t nt64 = (d->cout_pad + 63) / 64, nt32 = (d->cout_pad + 31) / 32;
  int sel = g_small_force;
  if (!sel) {
    if (mt64 * nt64 >= 384) sel = 1;
    else if (mt64 * nt32 >= 256) sel = 2;
    else if (mt32 * nt32 >= 192) sel = 3;
    else sel = 4;
  }
  hipStream_t st = (hipStream_t)stream;
  (void)mt16;
  switch (sel) {
    case 1: return launch<4, 4, 4>(k, st);
    case 2: return launch<4, 2, 4>(k, st);
    case 3: return nk >= 32 ? launch<2, 2, 8>(k, st) : launch<2, 2, 4>(k, st);
    default: return nk >= 32 ? launch<1, 2, 8>(k, st) : launch<1, 2, 4>(k, st);
  }
}
