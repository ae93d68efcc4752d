// Weight gradient with LDS-staged spatial tiles (fp16), 1x1 and 3x3 (any dilation), stride 1 or 2.
//
//   dW[co][ci][t] += sum_{n,oy,ox} dy[n,oy,ox,co] * x[n, oy*s + dy_t, ox*s + dx_t, ci]
//
// The per-tap / fused kernels of conv_wgrad.hip stage a 32-pixel K step per barrier pair and load every shifted x tile of a 3x3
// separately (9 global loads and 9 LDS stores per pixel and step: r1 profile 92 us for a 64->64 layer whose operands stream in 6 us).
// Here a workgroup owns a [32*COF x 32*CIF] block of the gradient for ALL taps and walks TH x 32-pixel tiles of the feature map:
//   * waves 4..7 are LOADERS: they bring the dy tile and the x HALO tile ((TH-1)*s+kh x 31*s+kw pixels) of tile i+1 into the other
//     LDS buffer (8 independent 16-byte loads in flight per thread) while
//   * waves 0..3 (2 x 2 over co x ci) run the MFMAs of tile i: the K fragment is 32 consecutive pixels of one tile row; both operands
//     are pixel-major in LDS exactly as in HBM and are read with the gfx950 transpose read ds_read_b64_tr_b16 (4 consecutive K values
//     of the lane's own channel), the dy fragments of a row are read once and reused by all taps;
//   * one barrier per tile; accumulators (taps x COF x CIF fragments) stay in registers over the workgroup's whole tile range
//     (split-K over tiles: partial blocks go to the workspace of conv_wgrad.hip's reduce kernel, or fp32 atomics for 1x1).
// x is written to LDS once per tile (halo ratio ~1.4) instead of once per tap.
//
// wgrad_tile_dma_kernel (round 4): the same tiles, the same MFMA waves, but the loader waves fill a 3- or 4-stage ring by LDS-DMA
// (`global_load_lds_dwordx4`) instead of global -> registers -> ds_write into two buffers.  The register loaders had ONE batch of <= 8
// loads per thread in flight and waited for it twice per tile (dy, then x): a 54 KB tile cost two HBM round trips, ~5 us, against
// 0.3-1.2 us of MFMA work -- 34 us for a 1x1 128->128 layer whose operands stream in 3 us.  With LDS-DMA a loader wave issues its
// 1 KB pieces of tile i+2 / i+3 while the MFMA waves are on tile i; it waits with a COUNTED vmcnt (the younger tiles stay in flight
// across the raw s_barrier).  The LDS image keeps the padded pixel pitch of the transpose reads: a DMA piece is lane-linear in LDS, but
// every lane supplies its own SOURCE address -- the lanes whose 16-byte slot is padding (or a halo pixel outside the image, or a
// channel past the tensor) fetch the zero page.
#include "myolo_dev.h"
#include <stdlib.h>
#include <string.h>

namespace wgt {

constexpr int THREADS = 512;
constexpr int TW = 32;

struct WgT {
  const char* x; int64_t x_sn, x_sh, x_sw; int Hi, Wi, Cin;
  const char* dy; int64_t d_sn, d_sh, d_sw; int Ho, Wo, Cout, N;
  float* dw; float* ws;
  int ntaps, stride;
  int tap_off[9];                    // halo pixel offset of tap t: (dy_t - mindy) * hw + (dx_t - mindx)
  int mindy, mindx, hw, hh, TH;
  int tiles_x, tiles_y, ntiles, ksplit;
  int tiles_co, tiles_ci, cout_w, cin_w;
  int x_bytes, d_bytes;
  int dbuf_bytes, xbuf_bytes;        // bytes of one dy / x LDS buffer
  int pd, px;                        // LDS pixel-row pitches (bytes) of the dy / x tiles
  int nst, dpieces, xpieces, stage_bytes;   // wgrad_tile_dma_kernel: ring stages, 1 KB pieces of the dy / x area, (dpieces + xpieces) * 1024
  int out_tiles;                     // gradient blocks per pixel split: workgroup id -> (block, split), see block_of()
  int nt;                            // LDS-DMA with the non-temporal policy (aux = 2): operands this launch reads once
  int dbg;                           // profiling only (myolo_set_option("wgrad_tile_dbg", bits)): 1 no LDS-DMA, 2 no fragment reads / MFMAs, 4 no result stores
};


// Workgroup id -> (gradient block b, pixel split): linear.  Every block of one split reads the SAME dy / x pixels (x once per co block, dy
// once per ci block: 2-8 x the operand bytes for the 128+-channel layers; PMC: 5.27 GB fetched per step for 3.25 GB of operands = exactly
// the issued bytes).  An XCD-aware order that put the blocks of a split on one XCD's L2 was built in round 4 and measured neutral standalone
// (762 vs 737 us over 16 layers, profiles/r4e_wgrad_xcd_ubench.txt) and in the step (7.845 vs 7.858 ms): removed in round 5.
__device__ __forceinline__ bool block_of(const WgT& p, int& b, int& split) {
  const int id = blockIdx.x;
  split = id / p.out_tiles;
  b = id % p.out_tiles;
  return true;
}

template <int NT, int COF, int CIF>
__global__ __launch_bounds__(THREADS) void wgrad_tile_kernel(const WgT p) {
  constexpr int CO_T = 32 * COF, CI_T = 32 * CIF;
  // LDS pixel-row pitches (bytes): an ODD number of 32-byte units per pixel step (dy and stride-1 x: row + 32; stride-2 x: row + 16, the
  // step is two rows).  A 32-lane group of a transpose read fetches 32-byte pieces of EIGHT CONSECUTIVE pixel steps (K index
  // 8g + 4h + k' <-> pixel 4g + k' + 16h, the same bijection for both operands): 8 x odd x 32 B covers the 64 banks exactly once at
  // every base alignment (the tap shifts).  With pixels 8g + k' + 4h and pitch row + 16 the group read pixels {0-3, 8-11} whose
  // pieces overlapped pairwise: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.33 (1x1) .. 0.50 (3x3), r3 PMC pass.
  const int PD = p.pd, PX = p.px;
  constexpr int DV = CO_T / 8, XV = CI_T / 8;                  // 16-byte vectors per pixel
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sD = smem;                                  // [2][TH*32][PD]
  char* sX = smem + 2 * p.dbuf_bytes;               // [2][hh*hw][PX]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b, split;
  if (!block_of(p, b, split)) return;
  const int tci = b % p.tiles_ci; const int tco = b / p.tiles_ci;
  const int co0 = tco * CO_T, ci0 = tci * CI_T;
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const int ntl = split < p.ntiles ? (p.ntiles - split + p.ksplit - 1) / p.ksplit : 0;    // tiles split, split+ksplit, ...

  if (wave >= 4) {
    // ------------------------------------------------------------------ loaders
    const int lt = tid - 256;
    constexpr int OOB = 0x7fff0000;
    constexpr int U = 8;       // (16-32 loads in flight per thread measured SLOWER: 55 -> 66 us on 32->32 3x3, 34 -> 39 us on 128->128 1x1)
    const __amdgpu_buffer_rsrc_t rbx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.dy), 0, p.d_bytes, 0x00020000);
    const int ndv = p.TH * TW * DV, nxv = p.hh * p.hw * XV;
    auto load_tile = [&](int tile, int buf) {
      const int n = tile / tiles_per_img; const int r0 = tile - n * tiles_per_img;
      const int ty = r0 / p.tiles_x, tx = r0 - ty * p.tiles_x;
      const int oy0 = ty * p.TH, ox0 = tx * TW;
      char* dD = sD + buf * p.dbuf_bytes;
      char* dX = sX + buf * p.xbuf_bytes;
      const int dbase = (n * (int)p.d_sn + oy0 * (int)p.d_sh + ox0 * (int)p.d_sw + co0) * 2;
      for (int base = 0; base < ndv; base += 256 * U) {
        uint4 tmp[U]; int dst[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int v = base + u * 256 + lt;
          const int pix = v / DV, seg = v - pix * DV;
          const int r = pix >> 5, c = pix & 31;
          const bool ok = v < ndv && oy0 + r < p.Ho && ox0 + c < p.Wo && co0 + seg * 8 < p.Cout;
          const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rbd, ok ? dbase + (r * (int)p.d_sh + c * (int)p.d_sw + seg * 8) * 2 : OOB, 0, 0);
          tmp[u] = uint4{q.x, q.y, q.z, q.w};
          dst[u] = v < ndv ? pix * PD + seg * 16 : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (dst[u] >= 0) *reinterpret_cast<uint4*>(dD + dst[u]) = tmp[u];
      }
      const int iy0 = oy0 * p.stride + p.mindy, ix0 = ox0 * p.stride + p.mindx;
      const int xbase = (n * (int)p.x_sn + iy0 * (int)p.x_sh + ix0 * (int)p.x_sw + ci0) * 2;
      for (int base = 0; base < nxv; base += 256 * U) {
        uint4 tmp[U]; int dst[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int v = base + u * 256 + lt;
          const int pix = v / XV, seg = v - pix * XV;
          const int py = pix / p.hw, px = pix - py * p.hw;
          const bool ok = v < nxv && (unsigned)(iy0 + py) < (unsigned)p.Hi && (unsigned)(ix0 + px) < (unsigned)p.Wi && ci0 + seg * 8 < p.Cin;
          const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rbx, ok ? xbase + (py * (int)p.x_sh + px * (int)p.x_sw + seg * 8) * 2 : OOB, 0, 0);
          tmp[u] = uint4{q.x, q.y, q.z, q.w};
          dst[u] = v < nxv ? pix * PX + seg * 16 : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (dst[u] >= 0) *reinterpret_cast<uint4*>(dX + dst[u]) = tmp[u];
      }
    };
    if (ntl > 0) load_tile(split, 0);
    __syncthreads();
    for (int i = 0; i < ntl; ++i) {
      if (i + 1 < ntl) load_tile(split + (i + 1) * p.ksplit, (i + 1) & 1);
      __syncthreads();
    }
    return;
  }

  // ------------------------------------------------------------------ MFMA waves (2 x 2 over co x ci)
  const int wr = wave >> 1, wc = wave & 1;
  f4_t acc[NT][COF][CIF];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < COF; ++i)
#pragma unroll
      for (int j = 0; j < CIF; ++j) acc[t][i][j] = f4_t{0.f, 0.f, 0.f, 0.f};
  // transpose-read lane map (see conv_wgrad.hip): within a 16-lane group lane (4*k'+q) supplies the address of pixel (4g + k') (+16h),
  // channels base+4q..4q+3, and receives the 4 consecutive pixels of channel base + (lane&15)
  const int g = lane >> 4, kq = (lane & 15) >> 2, q = lane & 3;
  const int dlane = (4 * g + kq) * PD + (wr * 16 * COF + q * 4) * 2;
  const int xlane = (4 * g + kq) * p.stride * PX + (wc * 16 * CIF + q * 4) * 2;
  __syncthreads();                                    // first tile staged
  for (int i = 0; i < ntl; ++i) {
    const char* bD = sD + (i & 1) * p.dbuf_bytes + dlane;
    const char* bX = sX + (i & 1) * p.xbuf_bytes + xlane;
    for (int r = 0; r < p.TH; ++r) {
      h8_t fa[COF];
#pragma unroll
      for (int f = 0; f < COF; ++f)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const fp16x4_t va = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
              (__attribute__((address_space(3))) fp16x4_t*)(bD + (r * TW + 16 * h) * PD + f * 32));
#pragma unroll
          for (int e = 0; e < 4; ++e) fa[f][4 * h + e] = (half_t)va[e];
        }
      const int rowoff = r * p.stride * p.hw;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        h8_t fb[CIF];
        const int poff = (rowoff + p.tap_off[t]) * PX;
#pragma unroll
        for (int f = 0; f < CIF; ++f)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const fp16x4_t vb = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (__attribute__((address_space(3))) fp16x4_t*)(bX + poff + 16 * h * p.stride * PX + f * 32));
#pragma unroll
            for (int e = 0; e < 4; ++e) fb[f][4 * h + e] = (half_t)vb[e];
          }
#pragma unroll
        for (int ii = 0; ii < COF; ++ii)
#pragma unroll
          for (int jj = 0; jj < CIF; ++jj)
            acc[t][ii][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ii], fb[jj], acc[t][ii][jj], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // acc[t][i][j][r] = D[row(co) = wr*16*COF + i*16 + 4*(lane>>4) + r][col(ci) = wc*16*CIF + j*16 + (lane&15)]
  const int CoP = p.tiles_co * CO_T, CiP = p.tiles_ci * CI_T;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < COF; ++i)
#pragma unroll
      for (int j = 0; j < CIF; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + wr * 16 * COF + i * 16 + 4 * (lane >> 4) + r;
          const int ci = ci0 + wc * 16 * CIF + j * 16 + (lane & 15);
          if (p.ws) p.ws[(((int64_t)(split * NT + t) * CoP) + co) * CiP + ci] = acc[t][i][j][r];
          else if (co < p.cout_w && ci < p.cin_w) atomicAdd(p.dw + ((int64_t)co * p.cin_w + ci) * NT + t, acc[t][i][j][r]);
        }
}

// ---- LDS-DMA loaders (see the file comment) -------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

// s_waitcnt takes an immediate: the count (pieces of the younger tiles, wave-uniform) picks the instruction.  A count past the table
// waits for everything (always safe: it only gives up overlap).
__device__ __forceinline__ void wait_vmcnt(int n) {
  switch (n) {
#define WGT_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    WGT_W(1) WGT_W(2) WGT_W(3) WGT_W(4) WGT_W(5) WGT_W(6) WGT_W(7) WGT_W(8) WGT_W(9) WGT_W(10) WGT_W(11) WGT_W(12)
    WGT_W(13) WGT_W(14) WGT_W(15) WGT_W(16) WGT_W(17) WGT_W(18) WGT_W(19) WGT_W(20) WGT_W(21) WGT_W(22) WGT_W(23) WGT_W(24)
    WGT_W(25) WGT_W(26) WGT_W(27) WGT_W(28) WGT_W(29) WGT_W(30) WGT_W(31) WGT_W(32) WGT_W(33) WGT_W(34) WGT_W(35) WGT_W(36)
#undef WGT_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

constexpr int DMA_MAXP = 10;         // 1 KB pieces of the dy area, and of the x area, per loader wave and tile (host: each area <= 4 * DMA_MAXP KB)

template <int NT, int COF, int CIF>
__global__ __launch_bounds__(THREADS) void wgrad_tile_dma_kernel(const WgT p) {
  constexpr int CO_T = 32 * COF, CI_T = 32 * CIF;
  const int PD = p.pd, PX = p.px;
  constexpr int DV = CO_T / 8, XV = CI_T / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [nst][dy area: dpieces KB | x area: xpieces KB]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b, split;
  if (!block_of(p, b, split)) return;
  const int tci = b % p.tiles_ci; const int tco = b / p.tiles_ci;
  const int co0 = tco * CO_T, ci0 = tci * CI_T;
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const int ntl = split < p.ntiles ? (p.ntiles - split + p.ksplit - 1) / p.ksplit : 0;    // tiles split, split+ksplit, ...
  const int nst = p.nst;

  if (wave >= 4) {
    // ------------------------------------------------------------------ loaders: wave lw owns pieces lw, lw + 4, ... of every stage
    const int lw = wave - 4;
    // per piece and lane, once: the byte offset of the lane's 16-byte slot relative to the tile's first pixel and its (row << 16 |
    // column) in the tile for the per-tile image-bounds test; padding slots, slots past the tile and channels past the tensor get row
    // and column 0x7fff: they fail every bounds test and fetch the zero page.  dy and x pieces in separate tables: no per-piece selects.
    const int ndw = (p.dpieces - lw + 3) >> 2, nxw = (p.xpieces - lw + 3) >> 2;     // this wave's dy / x pieces per tile
    int doff[DMA_MAXP], drc[DMA_MAXP], xoff[DMA_MAXP], xrc[DMA_MAXP];
#pragma unroll
    for (int j = 0; j < DMA_MAXP; ++j) {
      doff[j] = 0; drc[j] = 0x7fff7fff; xoff[j] = 0; xrc[j] = 0x7fff7fff;
      const int so = (lw + 4 * j) * 1024 + lane * 16;
      {
        const int pix = so / PD, seg = (so - pix * PD) >> 4;
        const int r = pix >> 5, c = pix & 31;
        if (j < ndw && pix < p.TH * TW && seg < DV && co0 + seg * 8 < p.Cout) {
          doff[j] = (r * (int)p.d_sh + c * (int)p.d_sw + seg * 8) * 2;
          drc[j] = (r << 16) | c;
        }
      }
      {
        const int pix = so / PX, seg = (so - pix * PX) >> 4;
        const int py = pix / p.hw, px = pix - py * p.hw;
        if (j < nxw && pix < p.hh * p.hw && seg < XV && ci0 + seg * 8 < p.Cin) {
          xoff[j] = (py * (int)p.x_sh + px * (int)p.x_sw + seg * 8) * 2;
          xrc[j] = (py << 16) | px;
        }
      }
    }
    const uint64_t zp = (uint64_t)(uintptr_t)zero_page(), dyp = (uint64_t)(uintptr_t)p.dy, xp = (uint64_t)(uintptr_t)p.x;
    const unsigned Ho = (unsigned)p.Ho, Wo = (unsigned)p.Wo, Hi = (unsigned)p.Hi, Wi = (unsigned)p.Wi;
    auto issue = [&](int tile, int stg) {
      if (p.dbg & 1) return;
      const int n = tile / tiles_per_img; const int r0 = tile - n * tiles_per_img;
      const int ty = r0 / p.tiles_x, tx = r0 - ty * p.tiles_x;
      const int oy0 = ty * p.TH, ox0 = tx * TW;
      const int iy0 = oy0 * p.stride + p.mindy, ix0 = ox0 * p.stride + p.mindx;
      const int dbase = (n * (int)p.d_sn + oy0 * (int)p.d_sh + ox0 * (int)p.d_sw + co0) * 2;
      const int xbase = (n * (int)p.x_sn + iy0 * (int)p.x_sh + ix0 * (int)p.x_sw + ci0) * 2;
      char* sd = smem + stg * p.stage_bytes + lw * 1024;
      char* sx = sd + p.dpieces * 1024;
      // (flat selects on integers: a `cond ? pointer : zero_page()` chain made hipcc branch around every address computation)
#pragma unroll
      for (int j = 0; j < DMA_MAXP; ++j)
        if (j < ndw) {
          const bool ok = ((unsigned)(oy0 + (drc[j] >> 16)) < Ho) & ((unsigned)(ox0 + (drc[j] & 0xffff)) < Wo);
          const uint64_t a = dyp + (uint64_t)(int64_t)(dbase + doff[j]);
          if (p.nt) __builtin_amdgcn_global_load_lds((gptr_t*)(uintptr_t)(ok ? a : zp), (lptr_t*)(sd + j * 4096), 16, 0, 2);
          else __builtin_amdgcn_global_load_lds((gptr_t*)(uintptr_t)(ok ? a : zp), (lptr_t*)(sd + j * 4096), 16, 0, 0);
        }
#pragma unroll
      for (int j = 0; j < DMA_MAXP; ++j)
        if (j < nxw) {
          const bool ok = ((unsigned)(iy0 + (xrc[j] >> 16)) < Hi) & ((unsigned)(ix0 + (xrc[j] & 0xffff)) < Wi);
          const uint64_t a = xp + (uint64_t)(int64_t)(xbase + xoff[j]);
          if (p.nt) __builtin_amdgcn_global_load_lds((gptr_t*)(uintptr_t)(ok ? a : zp), (lptr_t*)(sx + j * 4096), 16, 0, 2);
          else __builtin_amdgcn_global_load_lds((gptr_t*)(uintptr_t)(ok ? a : zp), (lptr_t*)(sx + j * 4096), 16, 0, 0);
        }
    };
    const int ppw = ndw + nxw;
    // ring: tile i lives in stage i % nst; tiles i+1 .. i+nst-2 are in flight while the MFMA waves are on tile i.  Barrier i says: every
    // loader's pieces of tile i have landed (counted wait in front of it) AND every MFMA wave is done with tile i-1, whose stage the
    // next issue overwrites.
    const int pro = ntl < nst - 1 ? ntl : nst - 1;
    for (int j = 0; j < pro; ++j) issue(split + j * p.ksplit, j);
    int stg = pro == nst ? 0 : pro;                  // stage of the next tile to issue (pro <= nst - 1)
    for (int i = 0; i < ntl; ++i) {
      int younger = ntl - 1 - i;
      if (younger > nst - 2) younger = nst - 2;
      if (p.dbg & 1) wait_vmcnt(0); else wait_vmcnt(younger * ppw);
      asm volatile("s_barrier" ::: "memory");
      const int nx = i + nst - 1;
      if (nx < ntl) {
        issue(split + nx * p.ksplit, stg);
        stg = stg + 1 == nst ? 0 : stg + 1;
      }
    }
    return;
  }

  // ------------------------------------------------------------------ MFMA waves (2 x 2 over co x ci): as in wgrad_tile_kernel
  const int wr = wave >> 1, wc = wave & 1;
  f4_t acc[NT][COF][CIF];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < COF; ++i)
#pragma unroll
      for (int j = 0; j < CIF; ++j) acc[t][i][j] = f4_t{0.f, 0.f, 0.f, 0.f};
  const int g = lane >> 4, kq = (lane & 15) >> 2, q = lane & 3;
  const int dlane = (4 * g + kq) * PD + (wr * 16 * COF + q * 4) * 2;
  const int xlane = p.dpieces * 1024 + (4 * g + kq) * p.stride * PX + (wc * 16 * CIF + q * 4) * 2;
  int stg = 0;
  for (int i = 0; i < ntl; ++i) {
    asm volatile("s_barrier" ::: "memory");           // tile i has landed (the loaders waited for their pieces in front of it)
    const char* bD = smem + stg * p.stage_bytes + dlane;
    const char* bX = smem + stg * p.stage_bytes + xlane;
    stg = stg + 1 == nst ? 0 : stg + 1;
    if (p.dbg & 2) continue;
    for (int r = 0; r < p.TH; ++r) {
      h8_t fa[COF];
#pragma unroll
      for (int f = 0; f < COF; ++f)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const fp16x4_t va = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
              (__attribute__((address_space(3))) fp16x4_t*)(bD + (r * TW + 16 * h) * PD + f * 32));
#pragma unroll
          for (int e = 0; e < 4; ++e) fa[f][4 * h + e] = (half_t)va[e];
        }
      const int rowoff = r * p.stride * p.hw;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        h8_t fb[CIF];
        const int poff = (rowoff + p.tap_off[t]) * PX;
#pragma unroll
        for (int f = 0; f < CIF; ++f)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const fp16x4_t vb = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (__attribute__((address_space(3))) fp16x4_t*)(bX + poff + 16 * h * p.stride * PX + f * 32));
#pragma unroll
            for (int e = 0; e < 4; ++e) fb[f][4 * h + e] = (half_t)vb[e];
          }
#pragma unroll
        for (int ii = 0; ii < COF; ++ii)
#pragma unroll
          for (int jj = 0; jj < CIF; ++jj)
            acc[t][ii][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ii], fb[jj], acc[t][ii][jj], 0, 0, 0);
      }
    }
  }

  if (p.dbg & 4) return;
  const int CoP = p.tiles_co * CO_T, CiP = p.tiles_ci * CI_T;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < COF; ++i)
#pragma unroll
      for (int j = 0; j < CIF; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + wr * 16 * COF + i * 16 + 4 * (lane >> 4) + r;
          const int ci = ci0 + wc * 16 * CIF + j * 16 + (lane & 15);
          if (p.ws) p.ws[(((int64_t)(split * NT + t) * CoP) + co) * CiP + ci] = acc[t][i][j][r];
          else if (co < p.cout_w && ci < p.cin_w) atomicAdd(p.dw + ((int64_t)co * p.cin_w + ci) * NT + t, acc[t][i][j][r]);
        }
}

inline int grid_blocks(const WgT& k) { return k.out_tiles * k.ksplit; }

template <int NT, int COF, int CIF>
int launch(const WgT& k, int out_tiles, int smem, hipStream_t st) {
  if (k.nst > 0) {
    auto kd = wgrad_tile_dma_kernel<NT, COF, CIF>;
    MYOLO_ENSURE_DYN_SMEM(kd, smem);
    hipLaunchKernelGGL(kd, dim3(grid_blocks(k)), dim3(THREADS), smem, st, k);
    MYOLO_CHECK_LAUNCH();
    return 0;
  }
  auto kern = wgrad_tile_kernel<NT, COF, CIF>;
  MYOLO_ENSURE_DYN_SMEM(kern, smem);
  hipLaunchKernelGGL(kern, dim3(grid_blocks(k)), dim3(THREADS), smem, st, k);
  MYOLO_CHECK_LAUNCH();
  return 0;
}
template <int NT>
int launch_nt(const WgT& k, int cof, int cif, int out_tiles, int smem, hipStream_t st) {
  if constexpr (NT == 1) {           // 1x1: one accumulator set, room for 128-wide blocks (x and dy are read once per 128 channels)
    if (cof == 4 && cif == 4) return launch<NT, 4, 4>(k, out_tiles, smem, st);
    if (cof == 4 && cif == 2) return launch<NT, 4, 2>(k, out_tiles, smem, st);
    if (cof == 2 && cif == 4) return launch<NT, 2, 4>(k, out_tiles, smem, st);
    if (cof == 4 && cif == 1) return launch<NT, 4, 1>(k, out_tiles, smem, st);
    if (cof == 1 && cif == 4) return launch<NT, 1, 4>(k, out_tiles, smem, st);
  }
  if (cof == 2 && cif == 2) return launch<NT, 2, 2>(k, out_tiles, smem, st);
  if (cof == 2 && cif == 1) return launch<NT, 2, 1>(k, out_tiles, smem, st);
  if (cof == 1 && cif == 2) return launch<NT, 1, 2>(k, out_tiles, smem, st);
  return launch<NT, 1, 1>(k, out_tiles, smem, st);
}

}  // namespace wgt

static int g_wgt_off = -1;
static int g_wgt_lds_kb = -1;      // LDS budget of a workgroup (KB)
static int g_wgt_dma = -1;         // 1: LDS-DMA loaders (wgrad_tile_dma_kernel), 0: the register loaders of round 2
static int g_wgt_nst = -1;         // ring stages of the LDS-DMA kernel: 3, 4, 0 = whichever keeps more bytes in flight
static int g_wgt_min_tiles = -1;   // split-K: at least this many tiles per workgroup
static int g_wgt_wg = -1;          // workgroups aimed at per layer
static int g_wgt_dbg = 0;
// (round 5, measured and removed: 1x1 split-K partials through the workspace + reduce launch instead of fp32 atomics -- 857.5 vs 865.9 us over the
//  16 layer shapes, 7.765 vs 7.773 ms per step, gpurun_out/next_ws1_ubench.txt of the round's first call; the XCD-aware order; the cache-policy
//  knob: non-temporal LDS-DMA for 1x1 layers with ONE gradient block -- every byte read once -- is simply what the kernel does)
int myolo_wgrad_tile_set(const char* name, int value) {
  if (!strcmp(name, "wgrad_tile_off")) { g_wgt_off = value; return 0; }
  if (!strcmp(name, "wgrad_tile_lds_kb")) { g_wgt_lds_kb = value; return 0; }
  if (!strcmp(name, "wgrad_tile_dma")) { g_wgt_dma = value; return 0; }
  if (!strcmp(name, "wgrad_tile_nst")) { g_wgt_nst = value; return 0; }
  if (!strcmp(name, "wgrad_tile_min_tiles")) { g_wgt_min_tiles = value; return 0; }
  if (!strcmp(name, "wgrad_tile_wg")) { g_wgt_wg = value; return 0; }
  if (!strcmp(name, "wgrad_tile_dbg")) { g_wgt_dbg = value; return 0; }
  return MYOLO_EINVAL;
}

// returns -1 when the layer does not qualify; `out_ks` / `out_cop` / `out_cip`: split count and padded block dims of the workspace
// slices (for conv_wgrad.hip's reduce launch) when d->ws is used
int myolo_wgrad_tile_try(const myolo_wgrad_desc* d, void* stream, int* out_ks, int* out_cop, int* out_cip, int* used_ws) {
  using namespace wgt;
  if (g_wgt_off < 0) g_wgt_off = getenv("MYOLO_NO_WGRAD_TILE") != nullptr;
  if (g_wgt_off || d->x.dtype != MYOLO_F16 || d->db || d->up_shift != 0) return -1;
  if (d->ntaps != 1 && d->ntaps != 9) return -1;
  if (d->stride != 1 && d->stride != 2) return -1;
  const int cout_w = d->cout > 0 ? d->cout : d->dy.c, cin_w = d->cin > 0 ? d->cin : d->x.c;
  if (cout_w > d->dy.c || cin_w > d->x.c) return MYOLO_EINVAL;
  int mindy = 0, maxdy = 0, mindx = 0, maxdx = 0;
  for (int t = 0; t < d->ntaps; ++t) {
    mindy = d->tap_dy[t] < mindy ? d->tap_dy[t] : mindy; maxdy = d->tap_dy[t] > maxdy ? d->tap_dy[t] : maxdy;
    mindx = d->tap_dx[t] < mindx ? d->tap_dx[t] : mindx; maxdx = d->tap_dx[t] > maxdx ? d->tap_dx[t] : maxdx;
  }
  const int s = d->stride;
  const int big = d->ntaps == 1 ? 4 : 2;          // widest block per workgroup: 128 channels for 1x1, 64 for 3x3 (9 accumulator sets)
  const int cof = cout_w > 64 ? big : (cout_w > 32 ? 2 : 1), cif = cin_w > 64 ? big : (cin_w > 32 ? 2 : 1);
  const int CO_T = 32 * cof, CI_T = 32 * cif;
  const int PD = CO_T * 2 + 32, PX = CI_T * 2 + (s == 1 ? 32 : 16);      // odd 32-byte units per pixel STEP (kernel comment)
  const int hw = (TW - 1) * s + 1 + (maxdx - mindx);
  // tallest tile whose two buffer pairs fit in LDS (<= 144 KB), at most 8 rows and not taller than the map
  int th = 0, hh = 0, smem = 0, dbuf = 0, xbuf = 0;
  // (MYOLO_WGRAD_TILE_LDS_KB: a smaller budget leaves LDS for a main-stream workgroup on the same CU -- the weight gradients run BESIDE the
  // dgrad / BatchNorm chain; A/B knob)
  if (g_wgt_lds_kb < 0) g_wgt_lds_kb = 144;            // (round 4: 96 / 64 KB measured 7.79 / 7.81 ms against 7.77, 40 KB 10.0; myolo_set_option("wgrad_tile_lds_kb", n))
  const int lds_cap = g_wgt_lds_kb * 1024;
  if (g_wgt_dma < 0) g_wgt_dma = getenv("MYOLO_WGRAD_TILE_DMA") ? atoi(getenv("MYOLO_WGRAD_TILE_DMA")) : 1;
  if (g_wgt_nst < 0) g_wgt_nst = 0;                     // (myolo_set_option("wgrad_tile_nst" / "_wg" / "_min_tiles"): tests and sweeps)
  int nst = 0;
  if (g_wgt_dma) {
    // LDS-DMA ring: each area rounded up to whole 1 KB pieces; the tallest tile of a 3- and of a 4-stage ring, then the ring that
    // keeps more tile rows in LDS (ties: 4 stages for 1x1 -- shorter tiles cost nothing there --, 3 for k x k: a taller tile re-reads
    // less x halo)
    int bt[2] = {0, 0};
    for (int c = 0; c < 2; ++c) {
      const int ns = 3 + c;
      if (g_wgt_nst && g_wgt_nst != ns) continue;
      for (int t = 8; t >= 1; --t) {
        if (t > d->dy.h && t > 1) continue;
        const int h2 = (t - 1) * s + 1 + (maxdy - mindy);
        const int db_ = (t * TW * PD + 1023) / 1024 * 1024, xb_ = (h2 * hw * PX + 1023) / 1024 * 1024;
        if (ns * (db_ + xb_) <= lds_cap && db_ / 1024 <= 4 * DMA_MAXP && xb_ / 1024 <= 4 * DMA_MAXP) { bt[c] = t; break; }
      }
    }
    if (bt[0] || bt[1]) {
      if (bt[1] * 4 > bt[0] * 3 || (bt[1] * 4 == bt[0] * 3 && d->ntaps == 1)) { nst = 4; th = bt[1]; }
      else { nst = 3; th = bt[0]; }
      hh = (th - 1) * s + 1 + (maxdy - mindy);
      dbuf = (th * TW * PD + 1023) / 1024 * 1024; xbuf = (hh * hw * PX + 1023) / 1024 * 1024;
      smem = nst * (dbuf + xbuf);
    }
  }
  if (!nst) {
    for (int t = 8; t >= 1; --t) {
      if (t > d->dy.h && t > 1) continue;
      const int h2 = (t - 1) * s + 1 + (maxdy - mindy);
      const int db_ = (t * TW * PD + 15) / 16 * 16, xb_ = (h2 * hw * PX + 15) / 16 * 16;
      const int sm = 2 * (db_ + xb_);
      if (sm <= lds_cap) { th = t; hh = h2; smem = sm; dbuf = db_; xbuf = xb_; break; }
    }
  }
  if (!th) return -1;
  WgT k;
  k.x = (const char*)d->x.ptr; k.x_sn = d->x.sn; k.x_sh = d->x.sh; k.x_sw = d->x.sw; k.Hi = d->x.h; k.Wi = d->x.w; k.Cin = d->x.c;
  k.dy = (const char*)d->dy.ptr; k.d_sn = d->dy.sn; k.d_sh = d->dy.sh; k.d_sw = d->dy.sw;
  k.Ho = d->dy.h; k.Wo = d->dy.w; k.Cout = d->dy.c; k.N = d->dy.n;
  k.dw = d->dw; k.ntaps = d->ntaps; k.stride = s;
  k.mindy = mindy; k.mindx = mindx; k.hw = hw; k.hh = hh; k.TH = th;
  for (int t = 0; t < 9; ++t) k.tap_off[t] = t < d->ntaps ? (d->tap_dy[t] - mindy) * hw + (d->tap_dx[t] - mindx) : 0;
  k.tiles_x = (k.Wo + TW - 1) / TW; k.tiles_y = (k.Ho + th - 1) / th;
  const int64_t nt = (int64_t)k.N * k.tiles_x * k.tiles_y;
  if (nt <= 0 || nt > 0x3fffffff) return -1;
  k.ntiles = (int)nt;
  k.cout_w = cout_w; k.cin_w = cin_w;
  k.tiles_co = (cout_w + CO_T - 1) / CO_T; k.tiles_ci = (cin_w + CI_T - 1) / CI_T;
  const int out_tiles = k.tiles_co * k.tiles_ci;
  auto span = [&](const myolo_tensor& t) -> int64_t {
    return (((int64_t)t.n - 1) * t.sn + ((int64_t)t.h - 1) * t.sh + ((int64_t)t.w - 1) * t.sw + t.c) * 2;
  };
  const int64_t xb = span(d->x), db = span(d->dy);
  if (xb >= 0x3ffe0000LL || db >= 0x3ffe0000LL) return -1;
  k.x_bytes = (int)xb; k.d_bytes = (int)db;
  k.dbuf_bytes = dbuf; k.xbuf_bytes = xbuf; k.pd = PD; k.px = PX;
  k.dbg = g_wgt_dbg;
  k.nst = nst; k.dpieces = nst ? dbuf / 1024 : 0; k.xpieces = nst ? xbuf / 1024 : 0; k.stage_bytes = nst ? dbuf + xbuf : 0;
  // split-K over tiles: one workgroup per CU at most (the buffers take > 80 KB), at least ~6 tiles per workgroup
  if (g_wgt_wg < 0) g_wgt_wg = 128;
  if (g_wgt_min_tiles < 0) g_wgt_min_tiles = 6;
  const int mt = g_wgt_min_tiles > 0 ? g_wgt_min_tiles : 1;
  const int want = d->wg_hint > 0 ? d->wg_hint : g_wgt_wg;
  int ks = d->ksplit > 0 ? d->ksplit : (want + out_tiles - 1) / out_tiles;
  const int max_ks = (k.ntiles + mt - 1) / mt;
  if (ks > max_ks) ks = max_ks;
  if (ks < 1) ks = 1;
  const int CoP = k.tiles_co * CO_T, CiP = k.tiles_ci * CI_T;
  const int64_t slice_bytes = (int64_t)k.ntaps * CoP * CiP * sizeof(float);
  k.ws = nullptr;
  if (d->ws && k.ntaps > 1 && d->ws_bytes >= slice_bytes * 2 && ks > 1 && (((uintptr_t)d->ws) & 15) == 0) {
    const int64_t fit = d->ws_bytes / slice_bytes;
    if (ks > fit) ks = (int)fit;
    k.ws = d->ws;
  }
  k.ksplit = ks;
  k.out_tiles = out_tiles;
  k.nt = out_tiles == 1 && k.ntaps == 1;
  *out_ks = ks; *out_cop = CoP; *out_cip = CiP; *used_ws = k.ws != nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (k.ntaps == 9) return launch_nt<9>(k, cof, cif, out_tiles, smem, st);
  return launch_nt<1>(k, cof, cif, out_tiles, smem, st);
}
