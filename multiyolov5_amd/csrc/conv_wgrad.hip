// Weight gradient of the NHWC convolution on gfx950 MFMA.
//
//   dW[co][ci][t] += sum_{pixels m} dy[m][co] * x[m shifted by tap t][ci]        (fp32, OIHW = Parameter.grad layout)
//
// GEMM view per tap: out [Cout x Cin], contraction K = pixels.  Both operands are pixel-major in HBM
// (NHWC), i.e. K is the *slow* index of both -- the MFMA wants K-contiguous fragments.  The tile
// [32 pixels][64 channels] is therefore staged row-major in LDS exactly as it lies in HBM (16-byte
// coalesced loads, no transposition pass) and fragments are read with the gfx950 LDS transpose read
// `ds_read_b64_tr_b16` (fp16), which hands each lane 4 consecutive K values of its own column; the fp32
// parity path reads one dword per lane per 16x16x4 MFMA, which needs no transposition at all.
//
// One workgroup (256 threads, 4 waves as 2x2) owns a 64(co) x 64(ci) tile of one tap and a contiguous
// range of pixels (split-K); partial tiles are combined with fp32 atomics.  Replaces the autograd wgrad of
// nn.Conv2d (reference models/common.py:38,42; invoked by loss.backward() at train.py:371,392).
#include "myolo_dev.h"
#include <stdlib.h>

namespace {

constexpr int WT = 64;        // tile edge (channels)
constexpr int KP = 32;        // pixels per K step
constexpr int THREADS = 256;

struct WgradK {
  const char* x; int64_t x_sn, x_sh, x_sw; int Hi, Wi, Cin;
  const char* dy; int64_t d_sn, d_sh, d_sw; int Ho, Wo, Cout, N;
  float* dw; float* db;
  int ntaps, stride, up;
  int tap_dy[MYOLO_MAX_TAPS], tap_dx[MYOLO_MAX_TAPS];
  int M, ksplit, pix_per_split, tiles_co, tiles_ci;
  int cout_w, cin_w;   // real (unpadded) weight dims: bounds of dw
  int x_bytes, d_bytes, dense;   // buffer-descriptor spans; dense: 1x1 stride-1 over pixel-dense views (pixel m at m*sw)
  float* ws;           // split-K partials [ksplit][ntaps][tiles_co*64][tiles_ci*64] (plain stores) or NULL (atomics)
  int dbg;
  int sp_tx, sp_ty;    // wgrad_small_halo_kernel: 8 x 16 pixel tiles per row / column of the map (M = tiles in all, pix_per_split = tiles per split)
};

template <typename T> struct Pitch;                       // LDS row pitch in bytes for a [KP][64] tile
template <> struct Pitch<half_t> { static constexpr int V = 64 * 2 + 16; };
template <> struct Pitch<float> { static constexpr int V = 64 * 4 + 16; };

template <typename T>
__global__ __launch_bounds__(THREADS) void wgrad_kernel(const WgradK p) {
  constexpr int ES = (int)sizeof(T);
  constexpr int SEG = ET<T>::SEG;
  constexpr int PITCH = Pitch<T>::V;
  constexpr int SEGS_PER_ROW = WT / SEG;                  // 8 (f16) / 16 (f32)
  constexpr int LOADS = KP * SEGS_PER_ROW / THREADS;      // 1 (f16) / 2 (f32) 16-byte loads per thread per tile
  __shared__ __attribute__((aligned(16))) char sD[2][KP * PITCH];   // dy tile  [pixel][co]
  __shared__ __attribute__((aligned(16))) char sX[2][KP * PITCH];   // x tile   [pixel][ci]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;               // wave tile: co rows [32wr,+32), ci cols [32wc,+32)
  int b = blockIdx.x;
  const int tci = b % p.tiles_ci; b /= p.tiles_ci;
  const int tco = b % p.tiles_co; b /= p.tiles_co;
  const int tap = b;
  const int split = blockIdx.y;
  const int co0 = tco * WT, ci0 = tci * WT;
  const int HWo = p.Ho * p.Wo;
  const int Hlog = p.Hi << p.up, Wlog = p.Wi << p.up;
  const int tdy = p.tap_dy[tap], tdx = p.tap_dx[tap];
  const int m_begin = split * p.pix_per_split;
  int m_end = m_begin + p.pix_per_split;
  if (m_end > p.M) m_end = p.M;
  const int nsteps = (m_end - m_begin + KP - 1) / KP;

  f4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};
  float bias_part = 0.f;   // db partial: thread (tid<64) sums dy column co0+tid ; only for tci==0 && tap==0

  // buffer descriptors: out-of-range offsets load zeros (rows past the split, outside the image, padded channels): no branches,
  // 32-bit address arithmetic
  constexpr int OOB = 0x7fff0000;
  const __amdgpu_buffer_rsrc_t rbx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rbd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.dy), 0, p.d_bytes, 0x00020000);
  auto bl = [&](const __amdgpu_buffer_rsrc_t& r, int off) -> uint4 {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return uint4{v.x, v.y, v.z, v.w};
  };
  uint4 rd[LOADS], rx[LOADS];
  auto issue = [&](int s) {
#pragma unroll
    for (int l = 0; l < LOADS; ++l) {
      const int v = tid + l * THREADS;
      const int prow = v / SEGS_PER_ROW, cs = v - prow * SEGS_PER_ROW;
      const int m = m_begin + s * KP + prow;
      const bool live = m < m_end && !(p.dbg & 2);
      const int cd = co0 + cs * SEG, cx = ci0 + cs * SEG;
      int doff, xoff;
      if (p.dense) {                                   // 1x1 stride 1: no div/mod (it cost ~60 instructions per 4 MFMAs)
        doff = m * (int)p.d_sw + cd;
        xoff = m * (int)p.x_sw + cx;
        doff = (live && cd < p.Cout) ? doff * ES : OOB;
        xoff = (live && cx < p.Cin) ? xoff * ES : OOB;
      } else {
        const int mm = live ? m : m_begin;
        const int n = mm / HWo; const int rem = mm - n * HWo; const int oy = rem / p.Wo; const int ox = rem - oy * p.Wo;
        doff = (live && cd < p.Cout) ? (n * (int)p.d_sn + oy * (int)p.d_sh + ox * (int)p.d_sw + cd) * ES : OOB;
        int iy = oy * p.stride + tdy, ix = ox * p.stride + tdx;
        const bool xin = live && (unsigned)iy < (unsigned)Hlog && (unsigned)ix < (unsigned)Wlog && cx < p.Cin;
        iy >>= p.up; ix >>= p.up;
        xoff = xin ? (n * (int)p.x_sn + iy * (int)p.x_sh + ix * (int)p.x_sw + cx) * ES : OOB;
      }
      rd[l] = bl(rbd, doff);
      rx[l] = bl(rbx, xoff);
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int l = 0; l < LOADS; ++l) {
      const int v = tid + l * THREADS;
      const int prow = v / SEGS_PER_ROW, cs = v - prow * SEGS_PER_ROW;
      *reinterpret_cast<uint4*>(&sD[buf][prow * PITCH + cs * 16]) = rd[l];
      *reinterpret_cast<uint4*>(&sX[buf][prow * PITCH + cs * 16]) = rx[l];
    }
  };

  if (nsteps > 0) {
    issue(0);
    stage(0);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    if (s + 1 < nsteps) issue(s + 1);
    if constexpr (sizeof(T) == 2) {
      // A fragment (co): lane (i = lane&15, g = lane>>4) needs dy[pixel 8g..8g+7][co = base + i]:
      // two transpose reads of a [4 pixel][16 co] block; within the 16-lane group lane (4*k'+q) supplies
      // the address of pixel row (8g + k') (+4), channels base+4q..4q+3, and receives column (lane&15).
      const int g = lane >> 4, kq = (lane & 15) >> 2, q = lane & 3;
      h8_t fa[2], fb[2];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int ca = (wr * 32 + f * 16 + q * 4) * 2;
        const int cb = (wc * 32 + f * 16 + q * 4) * 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int prow = 8 * g + 4 * h + kq;
          fp16x4_t va = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
              (__attribute__((address_space(3))) fp16x4_t*)(&sD[buf][prow * PITCH + ca]));
          fp16x4_t vb = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
              (__attribute__((address_space(3))) fp16x4_t*)(&sX[buf][prow * PITCH + cb]));
#pragma unroll
          for (int e = 0; e < 4; ++e) { fa[f][4 * h + e] = (half_t)va[e]; fb[f][4 * h + e] = (half_t)vb[e]; }
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    } else {
      const int i16 = lane & 15, g = lane >> 4;
#pragma unroll
      for (int kk = 0; kk < KP / 4; ++kk) {
        const int prow = kk * 4 + g;
        float a[2], bb[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          a[f] = *reinterpret_cast<const float*>(&sD[buf][prow * PITCH + (wr * 32 + f * 16 + i16) * 4]);
          bb[f] = *reinterpret_cast<const float*>(&sX[buf][prow * PITCH + (wc * 32 + f * 16 + i16) * 4]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bb[j], acc[i][j], 0, 0, 0);
      }
    }
    if (p.db && tci == 0 && tap == 0 && tid < WT) {
      for (int r = 0; r < KP; ++r) bias_part += (float)*reinterpret_cast<const T*>(&sD[buf][r * PITCH + tid * ES]);
    }
    if (s + 1 < nsteps) stage(buf ^ 1);
    __syncthreads();
  }

  // acc[i][j][r] = D[row(co) = wr*32+i*16+4*(lane>>4)+r][col(ci) = wc*32+j*16+(lane&15)]
  if (p.ws) {
    // split-K partial tile -> workspace with plain 64-byte-coalesced stores; wgrad_reduce_kernel sums the splits.
    // (fp32 atomics straight into OIHW cost 120 of 165 us on a 3x3 64->64 layer: 4M scattered read-modify-writes.)
    const int CoP = p.tiles_co * WT, CiP = p.tiles_ci * WT;
    float* wt = p.ws + ((int64_t)(split * p.ntaps + tap) * CoP) * CiP;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + wr * 32 + i * 16 + 4 * (lane >> 4) + r;
          const int ci = ci0 + wc * 32 + j * 16 + (lane & 15);
          wt[(int64_t)co * CiP + ci] = acc[i][j][r];
        }
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + wr * 32 + i * 16 + 4 * (lane >> 4) + r;
          const int ci = ci0 + wc * 32 + j * 16 + (lane & 15);
          if (co < p.cout_w && ci < p.cin_w && !(p.dbg & 1)) atomicAdd(p.dw + ((int64_t)co * p.cin_w + ci) * p.ntaps + tap, acc[i][j][r]);
        }
  }
  if (p.db && tci == 0 && tap == 0 && tid < WT && co0 + tid < p.cout_w) atomicAdd(p.db + co0 + tid, bias_part);
}


// ---- k x k convolutions (fp16): all NT taps in ONE workgroup ---------------------------------------------------------------
// The per-tap kernel above re-streams dy and x once per tap and issues only 4 MFMAs per wave between barriers.  Here the dy
// tile of a 32-pixel K step is staged once and multiplied against the NT shifted x tiles (the shifted pixels are L1/L2 hits):
// 4*NT MFMAs per wave per step, dy traffic / NT, NT accumulator sets (144 VGPRs for 3x3) held across the whole pixel range.
template <int NT>
__global__ __launch_bounds__(THREADS) void wgrad_fused_kernel(const WgradK p) {
  typedef half_t T;
  constexpr int ES = 2, SEG = 8;
  constexpr int PITCH = Pitch<T>::V;
  constexpr int SEGS_PER_ROW = WT / SEG;                  // 8: one 16-byte load per thread covers a [32 px][64 ch] tile
  __shared__ __attribute__((aligned(16))) char sD[KP * PITCH];
  __shared__ __attribute__((aligned(16))) char sX[NT][KP * PITCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  int b = blockIdx.x;
  const int tci = b % p.tiles_ci; const int tco = b / p.tiles_ci;
  const int split = blockIdx.y;
  const int co0 = tco * WT, ci0 = tci * WT;
  const int HWo = p.Ho * p.Wo;
  const int Hlog = p.Hi << p.up, Wlog = p.Wi << p.up;
  const int m_begin = split * p.pix_per_split;
  int m_end = m_begin + p.pix_per_split;
  if (m_end > p.M) m_end = p.M;
  const int nsteps = m_end > m_begin ? (m_end - m_begin + KP - 1) / KP : 0;

  f4_t acc[NT][2][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = f4_t{0.f, 0.f, 0.f, 0.f};
  float bias_part = 0.f;

  const int prow = tid / SEGS_PER_ROW, cs = tid - prow * SEGS_PER_ROW;
  const int cd = co0 + cs * SEG, cx = ci0 + cs * SEG;
  constexpr int OOB = 0x7fff0000;
  const __amdgpu_buffer_rsrc_t rbx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rbd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.dy), 0, p.d_bytes, 0x00020000);
  auto bl = [&](const __amdgpu_buffer_rsrc_t& r, int off) -> uint4 {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return uint4{v.x, v.y, v.z, v.w};
  };
  uint4 rd, rx[NT];
  auto issue = [&](int s) {
    const int m = m_begin + s * KP + prow;
    const bool live = m < m_end;
    const int mm = live ? m : m_begin;
    const int n = mm / HWo; const int rem = mm - n * HWo; const int oy = rem / p.Wo; const int ox = rem - oy * p.Wo;
    rd = bl(rbd, (live && cd < p.Cout) ? (n * (int)p.d_sn + oy * (int)p.d_sh + ox * (int)p.d_sw + cd) * ES : OOB);
    const int xn = n * (int)p.x_sn + cx;
    const bool cok = live && cx < p.Cin;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      int iy = oy * p.stride + p.tap_dy[t], ix = ox * p.stride + p.tap_dx[t];
      const bool ok = cok && (unsigned)iy < (unsigned)Hlog && (unsigned)ix < (unsigned)Wlog;
      iy >>= p.up; ix >>= p.up;
      rx[t] = bl(rbx, ok ? (xn + iy * (int)p.x_sh + ix * (int)p.x_sw) * ES : OOB);
    }
  };

  if (nsteps > 0) issue(0);
  for (int s = 0; s < nsteps; ++s) {
    *reinterpret_cast<uint4*>(&sD[prow * PITCH + cs * 16]) = rd;
#pragma unroll
    for (int t = 0; t < NT; ++t) *reinterpret_cast<uint4*>(&sX[t][prow * PITCH + cs * 16]) = rx[t];
    __syncthreads();
    if (s + 1 < nsteps) issue(s + 1);                    // next step's loads fly while this one computes
    // dy fragments (see wgrad_kernel for the transpose-read lane map)
    const int g = lane >> 4, kq = (lane & 15) >> 2, q = lane & 3;
    h8_t fa[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int ca = (wr * 32 + f * 16 + q * 4) * 2;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pr = 8 * g + 4 * h + kq;
        fp16x4_t va = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(&sD[pr * PITCH + ca]));
#pragma unroll
        for (int e = 0; e < 4; ++e) fa[f][4 * h + e] = (half_t)va[e];
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      h8_t fb[2];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int cb = (wc * 32 + f * 16 + q * 4) * 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int pr = 8 * g + 4 * h + kq;
          fp16x4_t vb = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(&sX[t][pr * PITCH + cb]));
#pragma unroll
          for (int e = 0; e < 4; ++e) fb[f][4 * h + e] = (half_t)vb[e];
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[t][i][j], 0, 0, 0);
    }
    if (p.db && tci == 0 && tid < WT)
      for (int r = 0; r < KP; ++r) bias_part += (float)*reinterpret_cast<const T*>(&sD[r * PITCH + tid * ES]);
    __syncthreads();
  }

  const int CoP = p.tiles_co * WT, CiP = p.tiles_ci * WT;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + wr * 32 + i * 16 + 4 * (lane >> 4) + r;
          const int ci = ci0 + wc * 32 + j * 16 + (lane & 15);
          if (p.ws) p.ws[(((int64_t)(split * NT + t) * CoP) + co) * CiP + ci] = acc[t][i][j][r];
          else if (co < p.cout_w && ci < p.cin_w) atomicAdd(p.dw + ((int64_t)co * p.cin_w + ci) * NT + t, acc[t][i][j][r]);
        }
  if (p.db && tci == 0 && tid < WT && co0 + tid < p.cout_w) atomicAdd(p.db + co0 + tid, bias_part);
}

// ---- 3x3 convolutions with <= 16 input and <= 32 output channels (Focus: 12 -> 32 on the full-resolution map, the LAST weight
// gradient of the step and therefore on its critical path).  The 64x64 tile of the kernels above would be 7/8 padding there;
// here the output tile is [32 co][16 ci] per tap and the four waves split the PIXELS of a 128-pixel K step instead (32 each):
// 18 MFMAs per wave per step, all useful, 11 useful 16-byte loads per thread per 128 pixels.  The waves' partial tiles meet
// in LDS; the workgroup writes one compact [9][32][16] slice of the split-K workspace.
constexpr int KPS = 128;
__global__ __launch_bounds__(THREADS) void wgrad_fused_small_kernel(const WgradK p) {
  typedef half_t T;
  constexpr int ES = 2, SEG = 8, NT = 9;
  constexpr int PD = 32 * ES + 16;                        // dy row pitch (bytes): 32 channels + pad
  constexpr int PX = 16 * ES + 16;                        // x row pitch: 16 channels + pad
  __shared__ __attribute__((aligned(16))) char sD[KPS * PD];
  __shared__ __attribute__((aligned(16))) char sX[NT][KPS * PX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int split = blockIdx.y;
  const int HWo = p.Ho * p.Wo;
  const int Hlog = p.Hi << p.up, Wlog = p.Wi << p.up;
  const int m_begin = split * p.pix_per_split;
  int m_end = m_begin + p.pix_per_split;
  if (m_end > p.M) m_end = p.M;
  const int nsteps = m_end > m_begin ? (m_end - m_begin + KPS - 1) / KPS : 0;

  f4_t acc[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[t][i] = f4_t{0.f, 0.f, 0.f, 0.f};

  constexpr int OOB = 0x7fff0000;
  const __amdgpu_buffer_rsrc_t rbx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rbd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.dy), 0, p.d_bytes, 0x00020000);
  auto bl = [&](const __amdgpu_buffer_rsrc_t& r, int off) -> uint4 {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return uint4{v.x, v.y, v.z, v.w};
  };
  // dy tile [128 px][4 segments]: two loads per thread; x tiles [128 px][2 segments] per tap: one load per thread and tap
  const int prow_x = tid >> 1, cx = (tid & 1) * SEG;
  uint4 rd[2], rx[NT];
  auto issue = [&](int s) {
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const int v = tid + l * THREADS;
      const int prow = v >> 2, cd = (v & 3) * SEG;
      const int m = m_begin + s * KPS + prow;
      const bool live = m < m_end;
      const int mm = live ? m : m_begin;
      const int n = mm / HWo; const int rem = mm - n * HWo; const int oy = rem / p.Wo; const int ox = rem - oy * p.Wo;
      rd[l] = bl(rbd, (live && cd < p.Cout) ? (n * (int)p.d_sn + oy * (int)p.d_sh + ox * (int)p.d_sw + cd) * ES : OOB);
    }
    const int m = m_begin + s * KPS + prow_x;
    const bool live = m < m_end;
    const int mm = live ? m : m_begin;
    const int n = mm / HWo; const int rem = mm - n * HWo; const int oy = rem / p.Wo; const int ox = rem - oy * p.Wo;
    const int xn = n * (int)p.x_sn + cx;
    const bool cok = live && cx < p.Cin;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      int iy = oy * p.stride + p.tap_dy[t], ix = ox * p.stride + p.tap_dx[t];
      const bool ok = cok && (unsigned)iy < (unsigned)Hlog && (unsigned)ix < (unsigned)Wlog;
      iy >>= p.up; ix >>= p.up;
      rx[t] = bl(rbx, ok ? (xn + iy * (int)p.x_sh + ix * (int)p.x_sw) * ES : OOB);
    }
  };

  if (nsteps > 0) issue(0);
  for (int s = 0; s < nsteps; ++s) {
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const int v = tid + l * THREADS;
      *reinterpret_cast<uint4*>(&sD[(v >> 2) * PD + (v & 3) * 16]) = rd[l];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) *reinterpret_cast<uint4*>(&sX[t][prow_x * PX + (tid & 1) * 16]) = rx[t];
    __syncthreads();
    if (s + 1 < nsteps) issue(s + 1);                    // next step's loads fly while this one computes
    const int g = lane >> 4, kq = (lane & 15) >> 2, q = lane & 3;       // transpose-read lane map: see wgrad_kernel
    h8_t fa[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int ca = (f * 16 + q * 4) * 2;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pr = wave * 32 + 8 * g + 4 * h + kq;
        fp16x4_t va = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(&sD[pr * PD + ca]));
#pragma unroll
        for (int e = 0; e < 4; ++e) fa[f][4 * h + e] = (half_t)va[e];
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      h8_t fb;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pr = wave * 32 + 8 * g + 4 * h + kq;
        fp16x4_t vb = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(&sX[t][pr * PX + q * 8]));
#pragma unroll
        for (int e = 0; e < 4; ++e) fb[4 * h + e] = (half_t)vb[e];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb, acc[t][i], 0, 0, 0);
    }
    __syncthreads();
  }
  // the four waves' partial [9][32][16] tiles meet in LDS (the staging area is free after the last barrier)
  float* red = reinterpret_cast<float*>(&sX[0][0]);
  for (int i = tid; i < NT * 32 * 16; i += THREADS) red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = i * 16 + 4 * (lane >> 4) + r, ci = lane & 15;
        atomicAdd(red + (t * 32 + co) * 16 + ci, acc[t][i][r]);
      }
  __syncthreads();
  float* dst = p.ws + (int64_t)split * (NT * 32 * 16);
  for (int i = tid; i < NT * 32 * 16; i += THREADS) dst[i] = red[i];
}

// ---- the same layer class on spatial tiles.  wgrad_fused_small_kernel fetches every shifted x tile of the 3x3 separately: 11 16-byte
// loads per thread and 128 pixels, 352 bytes per pixel through the texture path for 96 bytes of operands (Focus, alone on the chip
// at the end of the backward: 108 us for 201 MB).  Here a K step is an 8 x 16 pixel tile of the map: its dy tile and ONE 10 x 18
// x halo tile go to LDS (3.4 loads per thread and step, 141 bytes per pixel) and the nine taps read the halo at shifted pixel
// rows -- the transpose read takes a per-lane address, so a tap is a uniform byte offset on the lane's base.  Stride 1, taps within
// +-1, map dims multiples of 8 x 16 (others stay on the kernel above).
constexpr int FTH = 8, FTW = 16, FHH = FTH + 2, FHW = FTW + 2;
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void wgrad_small_halo_kernel(const WgradK p) {
  constexpr int ES = 2, SEG = 8, NT = 9;
  constexpr int PD = 32 * ES + 16;                        // dy row pitch (bytes): 32 channels + pad
  constexpr int PX = 16 * ES + 16;                        // x row pitch: 16 channels + pad
  constexpr int SD_BYTES = KPS * PD, SX_BYTES = FHH * FHW * PX;
  static_assert(SD_BYTES + SX_BYTES >= NT * 32 * 16 * 4, "the reduction tile reuses the staging area");
  __shared__ __attribute__((aligned(16))) char smem[SD_BYTES + SX_BYTES];
  char* sD = smem;
  char* sX = smem + SD_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int split = blockIdx.y;
  const int t_begin = split * p.pix_per_split;
  int t_end = t_begin + p.pix_per_split;
  if (t_end > p.M) t_end = p.M;
  const int nsteps = t_end > t_begin ? t_end - t_begin : 0;
  const int tiles_img = p.sp_tx * p.sp_ty;

  f4_t acc[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[t][i] = f4_t{0.f, 0.f, 0.f, 0.f};

  constexpr int OOB = 0x7fff0000;
  const __amdgpu_buffer_rsrc_t rbx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rbd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.dy), 0, p.d_bytes, 0x00020000);
  auto bl = [&](const __amdgpu_buffer_rsrc_t& r, int off) -> uint4 {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return uint4{v.x, v.y, v.z, v.w};
  };
  // per-thread constants of the two dy loads (tile pixel v >> 2, segment v & 3) and the two halo loads (halo pixel v >> 1, segment v & 1)
  int d_off[2], x_hy[2], x_hx[2];
  bool x_on[2];
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const int v = tid + l * THREADS;
    const int prow = v >> 2, cd = (v & 3) * SEG;
    d_off[l] = cd < p.Cout ? ((prow >> 4) * (int)p.d_sh + (prow & 15) * (int)p.d_sw + cd) * ES : -1;
    const int hp = v >> 1;
    x_on[l] = hp < FHH * FHW && (v & 1) * SEG < p.Cin;
    x_hy[l] = hp / FHW - 1; x_hx[l] = hp % FHW - 1;
  }
  // two tiles of loads in flight per thread (the MFMAs of a tile take a fraction of the HBM latency: with one tile ahead and two
  // workgroups per CU every step waited for its loads, 97 us; see the launch for the workgroup count)
  uint4 rd[2][2], rx[2][2];
  auto issue = [&](int s, const int b) {
    const int tl = t_begin + s;
    const int n = tl / tiles_img; const int r = tl - n * tiles_img; const int by = r / p.sp_tx; const int bx = r - by * p.sp_tx;
    const int oy0 = by * FTH, ox0 = bx * FTW;
    const int dbase = (n * (int)p.d_sn + oy0 * (int)p.d_sh + ox0 * (int)p.d_sw) * ES;
#pragma unroll
    for (int l = 0; l < 2; ++l) rd[b][l] = bl(rbd, d_off[l] >= 0 ? dbase + d_off[l] : OOB);
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const int iy = oy0 + x_hy[l], ix = ox0 + x_hx[l];
      const bool ok = x_on[l] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      rx[b][l] = bl(rbx, ok ? (n * (int)p.x_sn + iy * (int)p.x_sh + ix * (int)p.x_sw + ((tid + l * THREADS) & 1) * SEG) * ES : OOB);
    }
  };
  // fragment addresses: K index 8g + 4h + k' of the wave's 32 pixels (two tile rows of 16) <-> tile pixel wave*32 + 8g + 4h + k'
  const int g = lane >> 4, kq = (lane & 15) >> 2, q = lane & 3;
  const int pr0 = wave * 32 + 8 * g + kq;                  // h = 0; h = 1 is 4 pixels further on the same tile row
  const int hb0 = ((pr0 >> 4) + 1) * FHW + (pr0 & 15) + 1; // its halo pixel at tap (0, 0)
  int tap_b[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) tap_b[t] = (p.tap_dy[t] * FHW + p.tap_dx[t]) * PX;

  auto step = [&](int s, const int b) {
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const int v = tid + l * THREADS;
      *reinterpret_cast<uint4*>(&sD[(v >> 2) * PD + (v & 3) * 16]) = rd[b][l];
      if ((v >> 1) < FHH * FHW) *reinterpret_cast<uint4*>(&sX[(v >> 1) * PX + (v & 1) * 16]) = rx[b][l];
    }
    __syncthreads();
    if (s + 2 < nsteps) issue(s + 2, b);                 // (the registers just stored are free)
    h8_t fa[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int ca = (f * 16 + q * 4) * 2;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        fp16x4_t va = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(&sD[(pr0 + 4 * h) * PD + ca]));
#pragma unroll
        for (int e = 0; e < 4; ++e) fa[f][4 * h + e] = (half_t)va[e];
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      h8_t fb;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        fp16x4_t vb = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
            (__attribute__((address_space(3))) fp16x4_t*)(&sX[(hb0 + 4 * h) * PX + tap_b[t] + q * 8]));
#pragma unroll
        for (int e = 0; e < 4; ++e) fb[4 * h + e] = (half_t)vb[e];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb, acc[t][i], 0, 0, 0);
    }
    __syncthreads();
  };
  if (nsteps > 0) issue(0, 0);
  if (nsteps > 1) issue(1, 1);
  for (int s = 0; s < nsteps; s += 2) {
    step(s, 0);
    if (s + 1 < nsteps) step(s + 1, 1);
  }
  // the four waves' partial [9][32][16] tiles meet in LDS, one wave after the other on a lane-linear image [t][i][lane][r] (16-byte
  // accesses, no bank conflicts; fp32 LDS atomics on the [t][co][ci] image were 4-way conflicted, 72 per lane)
  f4_t* red = reinterpret_cast<f4_t*>(smem);
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          f4_t v = acc[t][i];
          if (w) { const f4_t o = red[(t * 2 + i) * 64 + lane]; v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3]; }
          red[(t * 2 + i) * 64 + lane] = v;
        }
    }
    __syncthreads();
  }
  const float* redf = reinterpret_cast<const float*>(smem);
  float* dst = p.ws + (int64_t)split * (NT * 32 * 16);
  for (int e = tid; e < NT * 32 * 16; e += THREADS) {      // slice layout [t][co][ci]: co = 16 i + 4 (lane >> 4) + r, ci = lane & 15
    const int ci = e & 15, co = (e >> 4) & 31, t = e >> 9;
    dst[e] = redf[(((t * 2 + (co >> 4)) * 64) + ((co & 15) >> 2) * 16 + ci) * 4 + (co & 3)];
  }
}

// dw[co][ci][t] += sum over splits of ws[s][t][co][ci].  A workgroup owns 64 consecutive weights (coalesced along ci); its 4
// waves take every 4th split and combine through LDS.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int ks, int ntaps,
                                                           int CoP, int CiP, int cout, int cin) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t total = (int64_t)ntaps * cout * cin;
  const int64_t slice = (int64_t)ntaps * CoP * CiP;
  for (int64_t base = (int64_t)blockIdx.x * 64; base < total; base += (int64_t)gridDim.x * 64) {
    const int64_t i = base + lane;
    float a = 0.f;
    int ci = 0, co = 0, t = 0;
    if (i < total) {
      ci = (int)(i % cin); co = (int)((i / cin) % cout); t = (int)(i / ((int64_t)cin * cout));
      const float* src = ws + ((int64_t)t * CoP + co) * CiP + ci;
      float a0 = 0.f, a1 = 0.f;
      const int st = 4 * gridDim.y;                         // gridDim.y > 1: the splits are dealt over y, partial sums meet in fp32 atomics
      int s = wave + 4 * blockIdx.y;
      for (; s + st < ks; s += 2 * st) { a0 += src[(int64_t)s * slice]; a1 += src[(int64_t)(s + st) * slice]; }
      for (; s < ks; s += st) a0 += src[(int64_t)s * slice];
      a = a0 + a1;
    }
    red[wave][lane] = a;
    __syncthreads();
    if (wave == 0 && i < total) {
      const float v = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
      float* o = dw + ((int64_t)co * cin + ci) * ntaps + t;
      if (gridDim.y > 1) atomicAdd(o, v); else *o += v;
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int myolo_conv_wgrad(const myolo_wgrad_desc* d, void* stream) {
  if (!d || !d->x.ptr || !d->dy.ptr || !d->dw) return MYOLO_EINVAL;
  const int dt = d->x.dtype;
  if (dt != d->dy.dtype || (dt != MYOLO_F16 && dt != MYOLO_F32)) return MYOLO_EINVAL;
  const int seg = dt == MYOLO_F16 ? 8 : 4;
  if (d->ntaps < 1 || d->ntaps > MYOLO_MAX_TAPS) return MYOLO_EINVAL;
  if (d->x.c % seg || d->x.sw % seg || d->x.sh % seg || d->x.sn % seg || ((uintptr_t)d->x.ptr & 15)) return MYOLO_EINVAL;
  if (d->dy.sw % seg || d->dy.sh % seg || d->dy.sn % seg || ((uintptr_t)d->dy.ptr & 15)) return MYOLO_EINVAL;
  if (dt == MYOLO_F16 && !((d->cin > 0 ? d->cin : d->x.c) <= 16 && d->ntaps == 9)) {      // (Focus-sized 3x3: the compact kernel below)
    int ks2 = 0, cop = 0, cip = 0, uws = 0;
    const int r = myolo_wgrad_tile_try(d, stream, &ks2, &cop, &cip, &uws);     // LDS-staged spatial tiles (conv_wgrad_tile.hip)
    if (r != -1) {
      if (r != 0) return r;
      if (uws) {
        const int cw = d->cout > 0 ? d->cout : d->dy.c, iw = d->cin > 0 ? d->cin : d->x.c;
        const int64_t total = (int64_t)d->ntaps * cw * iw;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for(total, 64, 4096)), dim3(256), 0, (hipStream_t)stream, d->ws, d->dw, ks2,
                           d->ntaps, cop, cip, cw, iw);
        MYOLO_CHECK_LAUNCH();
      }
      return 0;
    }
  }
  WgradK k;
  k.x = (const char*)d->x.ptr; k.x_sn = d->x.sn; k.x_sh = d->x.sh; k.x_sw = d->x.sw; k.Hi = d->x.h; k.Wi = d->x.w; k.Cin = d->x.c;
  k.dy = (const char*)d->dy.ptr; k.d_sn = d->dy.sn; k.d_sh = d->dy.sh; k.d_sw = d->dy.sw;
  k.Ho = d->dy.h; k.Wo = d->dy.w; k.Cout = d->dy.c; k.N = d->dy.n;
  k.dw = d->dw; k.db = d->db; k.ntaps = d->ntaps; k.stride = d->stride; k.up = d->up_shift;
  for (int i = 0; i < MYOLO_MAX_TAPS; ++i) { k.tap_dy[i] = d->tap_dy[i]; k.tap_dx[i] = d->tap_dx[i]; }
  const int64_t M = (int64_t)k.N * k.Ho * k.Wo;
  if (M <= 0 || M > 0x7fffffff) return MYOLO_EINVAL;
  k.M = (int)M;
  {
    auto span = [&](const myolo_tensor& t) -> int64_t {
      return (((int64_t)t.n - 1) * t.sn + ((int64_t)t.h - 1) * t.sh + ((int64_t)t.w - 1) * t.sw + t.c) * (dt == MYOLO_F16 ? 2 : 4);
    };
    const int64_t xb = span(d->x), db = span(d->dy);
    if (xb >= 0x7ffe0000LL || db >= 0x7ffe0000LL) return MYOLO_EINVAL;          // 32-bit buffer offsets
    k.x_bytes = (int)xb; k.d_bytes = (int)db;
    auto pdense = [](const myolo_tensor& t) { return t.sh == (int64_t)t.w * t.sw && t.sn == (int64_t)t.h * t.sh; };
    k.dense = d->ntaps == 1 && d->stride == 1 && d->up_shift == 0 && d->tap_dy[0] == 0 && d->tap_dx[0] == 0 &&
              d->x.h == d->dy.h && d->x.w == d->dy.w && pdense(d->x) && pdense(d->dy);
  }
  k.cout_w = d->cout > 0 ? d->cout : k.Cout;
  k.cin_w = d->cin > 0 ? d->cin : k.Cin;
  if (k.cout_w > k.Cout || k.cin_w > k.Cin) return MYOLO_EINVAL;
  k.tiles_co = (k.cout_w + WT - 1) / WT;
  k.tiles_ci = (k.cin_w + WT - 1) / WT;
  const bool fused = dt == MYOLO_F16 && k.ntaps == 9;        // every tap in one workgroup (wgrad_fused_kernel)
  const int out_tiles = k.tiles_co * k.tiles_ci * (fused ? 1 : k.ntaps);
  int ks = d->ksplit;
  if (ks <= 0) {
    constexpr int tgt_fused = 128;
    static const int tgt_tap = getenv("MYOLO_WGRAD_WG") ? atoi(getenv("MYOLO_WGRAD_WG")) : 256;
    // Few, long-lived workgroups: these kernels run on the side stream BESIDE the dgrad / BatchNorm chain, so they need not fill the
    // chip, and every extra split costs a [taps][64][64] fp32 partial tile (written + re-read, or 4096 atomics per tap): measured
    // on the yolov5s step 1024/512 workgroups -> 256/128: 11.29 -> 10.63 ms; 128/64: 11.26 ms (the tail of the last layers shows).
    ks = ((fused ? tgt_fused : tgt_tap) + out_tiles - 1) / out_tiles;
    const int max_ks = (int)((M + 8 * KP - 1) / (8 * KP));    // but at least 8 K-steps each
    if (ks > max_ks) ks = max_ks;
    if (ks < 1) ks = 1;
  }
  const int CoP = k.tiles_co * WT, CiP = k.tiles_ci * WT;
  const int64_t slice_bytes = (int64_t)k.ntaps * CoP * CiP * sizeof(float);
  k.ws = nullptr;
  // 1x1 convs keep the atomics: their OIHW gradient is contiguous along ci, so the atomics coalesce (22 us for 4M of them) and
  // beat a second pass over ~1000 tiny slices; k x k gradients are strided by the tap count and go through the workspace.
  if (d->ws && k.ntaps > 1 && d->ws_bytes >= slice_bytes * 2 && ks > 1 && (((uintptr_t)d->ws) & 15) == 0) {
    const int64_t fit = d->ws_bytes / slice_bytes;
    if (ks > fit) ks = (int)fit;
    k.ws = d->ws;
  }
  // Focus-sized 3x3 (<= 16 -> <= 32 channels): compact [9][32][16] slices, waves split the pixels (wgrad_fused_small_kernel)
  const bool small = fused && !d->db && k.cout_w <= 32 && k.cin_w <= 16 && d->ws && (((uintptr_t)d->ws) & 15) == 0 &&
                     d->ws_bytes >= (int64_t)2 * 9 * 32 * 16 * (int64_t)sizeof(float);
  if (small) {
    constexpr int small_ks = 768;
    ks = d->ksplit > 0 ? d->ksplit : small_ks;                // three workgroups per CU (halo kernel, 256x512 map: 256 -> 66 us, 512 -> 52, 768 -> 50, 1024 -> 62)
    const int max_ks = (int)((M + 4 * KPS - 1) / (4 * KPS));
    if (ks > max_ks) ks = max_ks;
    const int64_t fit = d->ws_bytes / ((int64_t)9 * 32 * 16 * sizeof(float));
    if (ks > fit) ks = (int)fit;
    if (ks < 1) ks = 1;
    k.ws = d->ws;
  }
  int pps = (int)((M + ks - 1) / ks);
  const int kstep = small ? KPS : KP;
  pps = (pps + kstep - 1) / kstep * kstep;
  ks = (int)((M + pps - 1) / pps);
  k.ksplit = ks; k.pix_per_split = pps;
  constexpr int dbg = 0;
  k.dbg = dbg;
  hipStream_t st = (hipStream_t)stream;
  if (small) {
    bool tiles = k.stride == 1 && k.up == 0 && k.Hi == k.Ho && k.Wi == k.Wo && k.Ho % FTH == 0 && k.Wo % FTW == 0;
    for (int t = 0; t < 9; ++t) tiles = tiles && k.tap_dy[t] >= -1 && k.tap_dy[t] <= 1 && k.tap_dx[t] >= -1 && k.tap_dx[t] <= 1;
    constexpr int no_tiles = 0;
    if (tiles && !no_tiles) {                                 // 8 x 16 pixel tiles, one x halo for all taps (wgrad_small_halo_kernel)
      k.sp_tx = k.Wo / FTW; k.sp_ty = k.Ho / FTH;
      const int ntiles = k.N * k.sp_tx * k.sp_ty;
      const int tps = (ntiles + ks - 1) / ks;
      ks = (ntiles + tps - 1) / tps;
      k.M = ntiles; k.pix_per_split = tps; k.ksplit = ks;
      hipLaunchKernelGGL(wgrad_small_halo_kernel, dim3(1, ks), dim3(THREADS), 0, st, k);
    } else {
      hipLaunchKernelGGL(wgrad_fused_small_kernel, dim3(1, ks), dim3(THREADS), 0, st, k);
    }
    const int64_t total = (int64_t)9 * k.cout_w * k.cin_w;
    // (54 workgroups summing 512 slices each took 21 us at the very end of the step: deal the slices over 8 groups)
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for(total, 64, 4096), ks >= 64 ? 8 : 1), dim3(256), 0, st, k.ws, k.dw, ks, 9, 32, 16,
                       k.cout_w, k.cin_w);
    MYOLO_CHECK_LAUNCH();
    return 0;
  }
  if (fused) hipLaunchKernelGGL(wgrad_fused_kernel<9>, dim3(out_tiles, ks), dim3(THREADS), 0, st, k);
  else if (dt == MYOLO_F16) hipLaunchKernelGGL(wgrad_kernel<half_t>, dim3(out_tiles, ks), dim3(THREADS), 0, st, k);
  else hipLaunchKernelGGL(wgrad_kernel<float>, dim3(out_tiles, ks), dim3(THREADS), 0, st, k);
  if (k.ws) {
    const int64_t total = (int64_t)k.ntaps * k.cout_w * k.cin_w;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for(total, 64, 4096)), dim3(256), 0, st, k.ws, k.dw, ks, k.ntaps, CoP, CiP,
                       k.cout_w, k.cin_w);
  }
  MYOLO_CHECK_LAUNCH();
  return 0;
}
