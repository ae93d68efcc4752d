// Bottleneck in ONE launch for the fused (eval) model: cv1 = 1x1 Conv + folded BatchNorm + SiLU feeding cv2 = 3x3 Conv + folded BatchNorm +
// SiLU (+ the shortcut add) -- reference models/common.py:95-105 `x + cv2(cv1(x))` with Conv.fuseforward (common.py:45-46) after
// Model.fuse() (yolo.py:339-347), call site detect.py:144.  BatchNorm is folded in eval, so nothing forces a kernel boundary between the two
// convolutions; cv1's output has one reader and never goes to HBM ("cross-layer fusion in eval").
//
//   workgroup = TH x 16 output pixels of one image x BN output channels.
//   1. the (TH+2) x 18 input pixels (all C channels) go to LDS ONCE by LDS-DMA (conv_midx.hip's halo tile: 16-byte segment s of halo pixel hp at
//      physical segment s ^ (hp & 7) / low four bits ^ (hp & 15); the DMA image is lane-linear, the swizzle is applied to the SOURCE address)
//   2. GEMM 1 over the halo pixels: t = act1(scale1 * (W1 . x) + shift1) for ALL C mid channels (NMT passes of BN channels, accumulators of
//      every pass live in registers), W1 streamed through the weight ring (counted vmcnt, one raw barrier per 128-byte K step)
//   3. barrier; the t tile OVERWRITES the x tile in LDS (same pitch: Bottleneck has c_ -> c_ -> c_ channels), zeros for the halo pixels
//      outside the image (cv2 pads ITS input with zeros: silu(shift1) != 0 there)
//   4. GEMM 2 = conv_midx's main loop over the t tile, W2 streamed; epilogue act2(scale2 * acc + shift2) + shortcut, 16-byte NHWC stores
//   Every workgroup of an N tile recomputes GEMM 1 for its halo: (TH+2)*18 / (TH*16) = 1.4 (TH 8) .. 1.7 (TH 4) x a ninth of the 3x3's work.
//
// The intermediate is rounded to fp16 exactly where the two-launch form stores it, so the results agree with myolo_conv(a); myolo_conv(b) up
// to the fp32 summation order inside a K loop.  Layers that do not qualify run as those two launches (myolo_conv_pair below).
#include "myolo_dev.h"
#include <string.h>
#include <stdlib.h>

namespace cpair {

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

struct PairK {
  const char* x; const char* w1; const char* w2; char* y; const char* res;
  const float* sc1; const float* sh1; const float* sc2; const float* sh2;
  int act1, act2;
  int x_sn, x_sh, x_sw, y_sn, y_sh, y_sw, r_sn, r_sh, r_sw;      // bytes
  int H, W, Cout, C;                              // map, output channels of cv2, channels of x and t
  int kchunks, w1row_bytes, w2row_bytes;          // C / 64
  int tiles_x, tiles_y, ntiles, tiles_per_xcd;
  int HW, HP, npieces;                            // halo: HW = 18 pixels per row, HP = (TH+2) * HW, 1 KB pieces covering HPpad pixels
  int pshift, segmask, halo_bytes;
  int tap_hoff[9], tap_woff[9];
};

template <int TH, int BN, int WP, int WC, int NST, int NMT>
__global__ __launch_bounds__(64 * WP * WC) void conv_pair_kernel(const PairK p) {
  constexpr int TW = 16, BM = TH * TW;
  constexpr int NT = 64 * WP * WC, NW = WP * WC;
  constexpr int PW = BM / WP, CW = BN / WC;
  constexpr int PF = PW / 16, CF = CW / 16;
  static_assert(CF % 2 == 0 && PW % 16 == 0, "wave tile");
  constexpr int HPPAD = ((TH + 2) * (TW + 2) + 15) / 16 * 16, NF1 = HPPAD / 16, PF1 = (NF1 + WP - 1) / WP;
  constexpr int RPI = NT / 8, WR = BN / RPI, WSTAGE = BN * 128;
  static_assert(WR >= 1 && BN % RPI == 0, "weight rows over the loader lanes");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wp = wave / WC, wc = wave % WC;
  const int lq = lane >> 4, l15 = lane & 15;
  const int tn = blockIdx.y;
  const int xcd = blockIdx.x & 7, bslot = blockIdx.x >> 3, bstride = gridDim.x >> 3;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t*)smem;
  const unsigned ring0 = lds0 + p.halo_bytes;
  char* ring = smem + p.halo_bytes;
  float* cst = reinterpret_cast<float*>(smem + p.halo_bytes + NST * WSTAGE);     // [sc1: C][sh1: C][sc2: BN][sh2: BN]
  const unsigned cst0 = ring0 + NST * WSTAGE;

  for (int c = tid; c < p.C; c += NT) { cst[c] = p.sc1 ? p.sc1[c] : 1.f; cst[p.C + c] = p.sh1 ? p.sh1[c] : 0.f; }
  for (int c = tid; c < BN; c += NT) {
    const int cg = tn * BN + c;
    cst[2 * p.C + c] = (p.sc2 && cg < p.Cout) ? p.sc2[cg] : 1.f;
    cst[2 * p.C + BN + c] = (p.sh2 && cg < p.Cout) ? p.sh2[cg] : 0.f;
  }
  __syncthreads();

  const int lrow = tid >> 3;
  const int lsg = (tid & 7) ^ ((lrow >> 1) & 7);
  int wrow[WR];                                   // ring row j * RPI + lrow of a stage holds channel panel_chan(row) of the N tile / mid pass
#pragma unroll
  for (int j = 0; j < WR; ++j) wrow[j] = panel_chan(j * RPI + lrow);
  unsigned foff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) foff[kk] = l15 * 128 + (((kk * 4 + lq) ^ ((l15 >> 1) & 7)) << 4);

  const int nseg = 1 << (p.pshift - 4);
  const int ppp = 1024 >> p.pshift;
  const int n1 = NMT * p.kchunks, n2 = 9 * p.kchunks;

  for (int tslot = bslot; tslot < p.tiles_per_xcd; tslot += bstride) {
    const int tile = xcd * p.tiles_per_xcd + tslot;
    if (tile >= p.ntiles) break;
    const int txy = p.tiles_x * p.tiles_y;
    const int n = tile / txy; const int trem = tile - n * txy;
    const int ty0 = (trem / p.tiles_x) * TH, tx0 = (trem - (trem / p.tiles_x) * p.tiles_x) * TW;

    // ---- x halo -> LDS (pixels past HP and outside the image: zero page) ----
    for (int pi = wave; pi < p.npieces; pi += NW) {
      const int hp = pi * ppp + (lane >> (p.pshift - 4));
      const int phys = lane & (nseg - 1);
      const int lseg = (phys & ~p.segmask) | ((phys ^ hp) & p.segmask);
      const int hy = hp / p.HW, hx = hp - hy * p.HW;
      const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
      const bool ok = hp < p.HP && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const char* src = ok ? p.x + (unsigned)(n * p.x_sn + iy * p.x_sh + ix * p.x_sw + lseg * 16) : zero_page();
      __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(smem + pi * 1024), 16, 0, 0);
    }

    const char* src[WR];
    auto piece = [&](int i, int buf) {
      __builtin_amdgcn_global_load_lds((gptr_t*)src[i], (lptr_t*)(ring + buf * WSTAGE + wave * 1024 + i * RPI * 128), 16, 0, 0);
    };

    // ================================ GEMM 1: t = act1(sc1 * W1 . x + sh1) over the halo pixels ================================
    f4_t acc1[NMT][CF][PF1];
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
      for (int c = 0; c < CF; ++c)
#pragma unroll
        for (int q = 0; q < PF1; ++q) acc1[m][c][q] = f4_t{0.f, 0.f, 0.f, 0.f};
    {
      int i_m = 0, i_kc = 0;                        // load cursor
      auto addresses = [&]() {
#pragma unroll
        for (int j = 0; j < WR; ++j) src[j] = p.w1 + (unsigned)((i_m * BN + wrow[j]) * p.w1row_bytes + lsg * 16 + i_kc * 128);
        if (++i_kc == p.kchunks) { i_kc = 0; ++i_m; }
        if (i_m >= NMT) i_m = NMT - 1;              // (addresses() may run once past the end: keep it inside the tensor)
      };
      int c_m = 0, c_kc = 0;                        // compute cursor
      auto step = [&](int buf, int nb, const bool loads) {
        const unsigned aw = ring0 + buf * WSTAGE + (wc * CW) * 128;
        unsigned ax[2][PF1];
#pragma unroll
        for (int q = 0; q < PF1; ++q) {
          const int hp = (wp + q * WP) * 16 + l15;
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const int seg = c_kc * 8 + kk * 4 + lq;
            ax[kk][q] = lds0 + ((unsigned)hp << p.pshift) + ((unsigned)((seg & ~p.segmask) | ((seg ^ hp) & p.segmask)) << 4);
          }
        }
        const int m_now = c_m;
        if (++c_kc == p.kchunks) { c_kc = 0; ++c_m; }
        u32x4_t wf[2][CF], xf[2][PF1];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
          for (int c = 0; c < CF; ++c) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[kk][c]) : "v"(aw + foff[kk]), "n"(c * 2048) : "memory");
#pragma unroll
          for (int q = 0; q < PF1; ++q) asm volatile("ds_read_b128 %0, %1" : "=v"(xf[kk][q]) : "v"(ax[kk][q]) : "memory");
        }
        if (loads) addresses();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          __builtin_amdgcn_sched_barrier(0);
          if (kk == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(CF + PF1) : "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int c = 0; c < CF; ++c) asm volatile("" : "+v"(wf[kk][c]));
#pragma unroll
          for (int q = 0; q < PF1; ++q) asm volatile("" : "+v"(xf[kk][q]));
#pragma unroll
          for (int m = 0; m < NMT; ++m)
            if (m == m_now) {
#pragma unroll
              for (int c = 0; c < CF; ++c)
#pragma unroll
                for (int q = 0; q < PF1; ++q)
                  if ((wp + q * WP) < NF1)
                    acc1[m][c][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const h8_t*>(&wf[kk][c]),
                                                                          *reinterpret_cast<const h8_t*>(&xf[kk][q]), acc1[m][c][q], 0, 0, 0);
            }
          if (loads && kk == 0) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < WR; ++i) piece(i, nb);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };
      const int inflight = n1 < NST - 1 ? n1 : NST - 1;
#pragma unroll
      for (int j = 0; j < NST - 1; ++j)
        if (j < inflight) {
          addresses();
#pragma unroll
          for (int i = 0; i < WR; ++i) piece(i, j);
        }
      int buf = 0;
      const int steady = n1 - (NST - 1);
      for (int s = 0; s < steady; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * WR) : "memory");
        __builtin_amdgcn_s_barrier();
        int nb = buf + NST - 1; nb = nb >= NST ? nb - NST : nb;
        step(buf, nb, true);
        buf = buf + 1 == NST ? 0 : buf + 1;
      }
#pragma unroll
      for (int r = NST - 2; r >= 0; --r) {
        if (r >= inflight) continue;
        if (r == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WR) : "memory");
        else if (r == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WR) : "memory");
        __builtin_amdgcn_s_barrier();
        step(buf, 0, false);
        buf = buf + 1 == NST ? 0 : buf + 1;
      }
    }
    __builtin_amdgcn_s_barrier();                  // every wave is done with the x tile and the ring: t may overwrite x, W2 may enter the ring

    // ================================ GEMM 2 (conv_midx.hip's loop over the t tile) ================================
    f4_t acc[CF][PF];
#pragma unroll
    for (int c = 0; c < CF; ++c)
#pragma unroll
      for (int q = 0; q < PF; ++q) acc[c][q] = f4_t{0.f, 0.f, 0.f, 0.f};
    int hbase[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) hbase[q] = (wp * PF + q) * p.HW + l15;
    int i_tap = 0, i_kc = 0;
    int n_wt = p.tap_woff[0];
    auto addresses2 = [&]() {
      const int wo = n_wt + i_kc * 128;
#pragma unroll
      for (int j = 0; j < WR; ++j) src[j] = p.w2 + (unsigned)((tn * BN + wrow[j]) * p.w2row_bytes + lsg * 16 + wo);
      if (++i_kc == p.kchunks) { i_kc = 0; ++i_tap; }
      n_wt = p.tap_woff[i_tap < 9 ? i_tap : 0];
    };
    const int inflight2 = n2 < NST - 1 ? n2 : NST - 1;
#pragma unroll
    for (int j = 0; j < NST - 1; ++j)
      if (j < inflight2) {                            // W2's first stages travel while the t tile is written
        addresses2();
#pragma unroll
        for (int i = 0; i < WR; ++i) piece(i, j);
      }

    // ---- epilogue 1: t -> LDS, in place of x ----
#pragma unroll
    for (int q = 0; q < PF1; ++q) {
      const int f = wp + q * WP;
      if (f < NF1) {
        const int hp = f * 16 + l15;
        const int hy = hp / p.HW, hx = hp - hy * p.HW;
        const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
        const bool ok = hp < p.HP && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
          for (int h = 0; h < CF / 2; ++h) {
            const int c0 = m * BN + wc * CW + 32 * h + 8 * lq;        // 8 consecutive mid channels of this lane
            u32x4_t kq[4];                                            // [0..1]: sc1[c0 .. c0+7], [2..3]: sh1[c0 .. c0+7]
            const unsigned ca = cst0 + c0 * 4;
            asm volatile("ds_read_b128 %0, %1" : "=v"(kq[0]) : "v"(ca) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(kq[1]) : "v"(ca) : "memory");
            asm volatile("ds_read_b128 %0, %1" : "=v"(kq[2]) : "v"(ca + (unsigned)p.C * 4) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(kq[3]) : "v"(ca + (unsigned)p.C * 4) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(kq[i]));
            // (element-wise __builtin_bit_cast(float, k.y) of an ext_vector read element x for every lane here: hipcc 7.2; go through memory
            //  like conv_mid.hip's apply constants)
            const float* sc = reinterpret_cast<const float*>(&kq[0]);
            const float* sh = reinterpret_cast<const float*>(&kq[2]);
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = acc1[m][2 * h][q][r]; v[4 + r] = acc1[m][2 * h + 1][q][r]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = ok ? act_f(fmaf(v[i], sc[i], sh[i]), p.act1) : 0.f;
            const u32x4_t o = pack_h8(v);
            const int seg = c0 >> 3;
            const unsigned ta = lds0 + ((unsigned)hp << p.pshift) + ((unsigned)((seg & ~p.segmask) | ((seg ^ hp) & p.segmask)) << 4);
            asm volatile("ds_write_b128 %0, %1" ::"v"(ta), "v"(o) : "memory");
          }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the t slots are in LDS before GEMM 2's first barrier lets the fragment reads go

    int c_tap = 0, c_kc = 0;
    int c_hoff = p.tap_hoff[0], n_hoff = p.tap_hoff[1];
    constexpr int G = 2 * CF;
    auto step2 = [&](int buf, int nb, const bool loads) {
      const unsigned aw = ring0 + buf * WSTAGE + (wc * CW) * 128;
      unsigned ax[2][PF];
#pragma unroll
      for (int q = 0; q < PF; ++q) {
        const int hp = hbase[q] + c_hoff;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int seg = c_kc * 8 + kk * 4 + lq;
          ax[kk][q] = lds0 + ((unsigned)hp << p.pshift) + ((unsigned)((seg & ~p.segmask) | ((seg ^ hp) & p.segmask)) << 4);
        }
      }
      if (++c_kc == p.kchunks) { c_kc = 0; ++c_tap; c_hoff = n_hoff; n_hoff = p.tap_hoff[c_tap + 1 < 9 ? c_tap + 1 : 0]; }
      u32x4_t wf[2][CF], xf[2][PF];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int c = 0; c < CF; ++c) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[kk][c]) : "v"(aw + foff[kk]), "n"(c * 2048) : "memory");
#pragma unroll
        for (int q = 0; q < PF; ++q) asm volatile("ds_read_b128 %0, %1" : "=v"(xf[kk][q]) : "v"(ax[kk][q]) : "memory");
      }
      if (loads) addresses2();
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        __builtin_amdgcn_sched_barrier(0);
        if (kk == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(CF + PF) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int c = 0; c < CF; ++c) asm volatile("" : "+v"(wf[kk][c]));
#pragma unroll
        for (int q = 0; q < PF; ++q) asm volatile("" : "+v"(xf[kk][q]));
#pragma unroll
        for (int c = 0; c < CF; ++c) {
#pragma unroll
          for (int q = 0; q < PF; ++q)
            acc[c][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const h8_t*>(&wf[kk][c]), *reinterpret_cast<const h8_t*>(&xf[kk][q]),
                                                               acc[c][q], 0, 0, 0);
          const int g = kk * CF + c;
#pragma unroll
          for (int i = 0; i < WR; ++i)
            if ((i * G) / WR == g && loads) {
              __builtin_amdgcn_sched_barrier(0);
              piece(i, nb);
              __builtin_amdgcn_sched_barrier(0);
            }
        }
      }
    };
    {
      int buf = 0;
      const int steady = n2 - (NST - 1);
      for (int s = 0; s < steady; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * WR) : "memory");
        __builtin_amdgcn_s_barrier();
        int nb = buf + NST - 1; nb = nb >= NST ? nb - NST : nb;
        step2(buf, nb, true);
        buf = buf + 1 == NST ? 0 : buf + 1;
      }
#pragma unroll
      for (int r = NST - 2; r >= 0; --r) {
        if (r >= inflight2) continue;
        if (r == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WR) : "memory");
        else if (r == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WR) : "memory");
        __builtin_amdgcn_s_barrier();
        step2(buf, 0, false);
        buf = buf + 1 == NST ? 0 : buf + 1;
      }
    }
    __builtin_amdgcn_s_barrier();                  // (the next tile's halo fill / W1 stages overwrite what the slowest wave may still be reading)

    // ---- epilogue 2: act2(sc2 * acc + sh2) + shortcut ----
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int oy = ty0 + wp * PF + q, ox = tx0 + l15;
      const bool mvalid = oy < p.H && ox < p.W;
      const unsigned yoff = (unsigned)(n * p.y_sn + oy * p.y_sh + ox * p.y_sw);
      const unsigned roff = (unsigned)(n * p.r_sn + oy * p.r_sh + ox * p.r_sw);
      uint4 rv[CF / 2];
#pragma unroll
      for (int h = 0; h < CF / 2; ++h) {
        const int c0 = tn * BN + wc * CW + 32 * h + 8 * lq;
        if (p.res) rv[h] = ldg16((mvalid && c0 < p.Cout) ? p.res + roff + c0 * 2 : zero_page());
      }
#pragma unroll
      for (int h = 0; h < CF / 2; ++h) {
        const int cl = wc * CW + 32 * h + 8 * lq;
        const int c0 = tn * BN + cl;
        const float* s2 = cst + 2 * p.C + cl;
        const float* h2 = s2 + BN;
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = acc[2 * h][q][r]; v[4 + r] = acc[2 * h + 1][q][r]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = act_f(fmaf(v[i], s2[i], h2[i]), p.act2);
        if (mvalid && c0 < p.Cout) {
          if (p.res) add_h8(v, u32x4_t{rv[h].x, rv[h].y, rv[h].z, rv[h].w});
          const u32x4_t o = pack_h8(v);
          stg16(p.y + yoff + c0 * 2, uint4{o.x, o.y, o.z, o.w});
        }
      }
    }
  }
}

template <int TH, int BN, int WP, int WC, int NST, int NMT>
int launch(const PairK& k, int per_cu, int ntile_c, hipStream_t st) {
  constexpr int NT = 64 * WP * WC;
  const int smem = k.halo_bytes + NST * BN * 128 + (2 * k.C + 2 * BN) * 4;
  if (smem > 160 * 1024) return -1;
  int fit = (160 * 1024) / (smem + 256);
  if (fit < per_cu) per_cu = fit < 1 ? 1 : fit;
  int per_xcd = (256 * per_cu / ntile_c + 7) / 8;
  if (per_xcd < 1) per_xcd = 1;
  if (per_xcd > k.tiles_per_xcd) per_xcd = k.tiles_per_xcd;
  auto kern = conv_pair_kernel<TH, BN, WP, WC, NST, NMT>;
  MYOLO_ENSURE_DYN_SMEM(kern, smem);
  hipLaunchKernelGGL(kern, dim3(per_xcd * 8, ntile_c), dim3(NT), smem, st, k);
  MYOLO_CHECK_LAUNCH();
  return 0;
}

}  // namespace cpair

static int g_pair_mode = -1;       // 0: always two launches, 1: fused where the layer pair qualifies (MYOLO_CONV_PAIR)
static int g_pair_th = 0;          // 0: from the tile count; 4 / 8 forced (tests)
int myolo_conv_pair_set(const char* name, int value) {
  if (!strcmp(name, "pair_mode")) { g_pair_mode = value; return 0; }
  if (!strcmp(name, "pair_th")) { g_pair_th = value; return 0; }
  return MYOLO_EINVAL;
}

// -1: the pair does not qualify
static int pair_try(const myolo_conv_desc* a, const myolo_conv_desc* b, void* stream) {
  using namespace cpair;
  if (g_pair_mode < 0) g_pair_mode = getenv("MYOLO_CONV_PAIR") ? atoi(getenv("MYOLO_CONV_PAIR")) : 1;
  if (!g_pair_mode) return -1;
  if (a->x.dtype != MYOLO_F16 || a->y.dtype != MYOLO_F16 || b->x.dtype != MYOLO_F16 || b->y.dtype != MYOLO_F16) return -1;
  // a: 1x1 stride 1, plain epilogue; its output IS b's input
  if (a->ntaps != 1 || a->tap_dy[0] || a->tap_dx[0] || a->stride != 1 || a->up_shift || a->det_no || a->res.ptr || a->accumulate || a->stats ||
      (a->bnb && a->nbnb)) return -1;
  if (a->y.ptr != b->x.ptr || a->y.n != b->x.n || a->y.h != b->x.h || a->y.w != b->x.w || a->y.c != b->x.c || a->y.sn != b->x.sn ||
      a->y.sh != b->x.sh || a->y.sw != b->x.sw) return -1;
  const int C = a->x.c;
  if ((C != 64 && C != 128 && C != 256) || a->cin_pad != C || a->y.c != C || a->cout_pad != C || b->cin_pad != C) return -1;
  if (a->x.n != a->y.n || a->x.h != a->y.h || a->x.w != a->y.w) return -1;
  // b: 3x3 stride 1 dilation 1 over the same map
  if (b->ntaps != 9 || b->stride != 1 || b->up_shift || b->det_no || b->accumulate || b->stats || (b->bnb && b->nbnb)) return -1;
  if (b->y.n != b->x.n || b->y.h != b->x.h || b->y.w != b->x.w || b->cout_pad % 64 || b->y.c % 8) return -1;
  if (b->res.ptr && (b->res.c < b->y.c || b->res.dtype != MYOLO_F16)) return -1;
  bool seen[9] = {false};
  for (int t = 0; t < 9; ++t) {
    const int dy = b->tap_dy[t], dx = b->tap_dx[t];
    if (dy < -1 || dy > 1 || dx < -1 || dx > 1 || seen[(dy + 1) * 3 + dx + 1]) return -1;
    seen[(dy + 1) * 3 + dx + 1] = true;
  }
  auto extent = [](const myolo_tensor& t) { return ((int64_t)t.n * t.sn + (int64_t)t.h * t.sh + (int64_t)t.w * t.sw + t.c) * 2; };
  if (extent(a->x) >= (1ll << 31) || extent(b->y) >= (1ll << 31) || (b->res.ptr && extent(b->res) >= (1ll << 31))) return -1;
  if ((int64_t)b->cout_pad * b->wtaps * C * 2 >= (1ll << 31)) return -1;
  if (((uintptr_t)a->x.ptr & 15) || ((uintptr_t)b->y.ptr & 15) || (a->x.sw % 8) || (b->y.sw % 8) || (b->res.ptr && (b->res.sw % 8))) return -1;
  // what the kernel's address arithmetic silently assumes -- one weight tap in `a` at slot 0 (w1row_bytes), 16-byte rows
  // everywhere (sn / sh multiples of 8 halves, an aligned residual of the OUTPUT's shape) -- and NO aliasing of the output with the input
  // halo or the residual: in the two-launch form an in-place Bottleneck is harmless (every element's residual is read before its output is
  // written by the same thread), here neighbouring workgroups still read the x halo while others store y
  if (a->wtaps != 1 || a->tap_w[0] != 0) return -1;
  if ((a->x.sn % 8) || (a->x.sh % 8) || (b->y.sn % 8) || (b->y.sh % 8)) return -1;
  if (b->res.ptr && (((uintptr_t)b->res.ptr & 15) || (b->res.sn % 8) || (b->res.sh % 8) || b->res.n != b->y.n || b->res.h != b->y.h || b->res.w != b->y.w))
    return -1;
  auto span = [](const myolo_tensor& t) {            // exact: one past the last byte the view touches
    return (uintptr_t)((((int64_t)t.n - 1) * t.sn + ((int64_t)t.h - 1) * t.sh + ((int64_t)t.w - 1) * t.sw + t.c) * 2);
  };
  auto overlap = [&](const myolo_tensor& p, const myolo_tensor& q) {
    const uintptr_t p0 = (uintptr_t)p.ptr, p1 = p0 + span(p), q0 = (uintptr_t)q.ptr, q1 = q0 + span(q);
    return p0 < q1 && q0 < p1;
  };
  // (channel slices of ONE concat buffer interleave in memory without sharing an element: their byte ranges overlap although the tensors
  //  do not -- the engine's plans are full of them.  Two views alias only if they also share channels: same pixel pitch and row walk, and the
  //  channel intervals [ptr, ptr + c) modulo the pixel pitch intersect)
  auto alias = [&](const myolo_tensor& p, const myolo_tensor& q) {
    if (!overlap(p, q)) return false;
    if (p.sw != q.sw || p.sh != q.sh || p.sn != q.sn) return true;                   // different walks over overlapping bytes: assume the worst
    const int64_t pitch = (int64_t)p.sw * 2;
    const int64_t d = (((int64_t)((uintptr_t)q.ptr - (uintptr_t)p.ptr)) % pitch + pitch) % pitch;   // q's first channel inside p's pixel, bytes
    const int64_t pc = (int64_t)p.c * 2, qc = (int64_t)q.c * 2;
    return d < pc || d + qc > pitch;                                                  // q starts inside p's channels, or wraps around into them
  };
  if (alias(a->x, b->y) || (b->res.ptr && alias(b->res, b->y) ) || alias(a->y, b->y)) return -1;
  PairK k;
  k.x = (const char*)a->x.ptr; k.w1 = (const char*)a->w; k.w2 = (const char*)b->w; k.y = (char*)b->y.ptr; k.res = (const char*)b->res.ptr;
  k.sc1 = a->scale; k.sh1 = a->shift; k.sc2 = b->scale; k.sh2 = b->shift; k.act1 = a->act; k.act2 = b->act;
  k.x_sn = (int)a->x.sn * 2; k.x_sh = (int)a->x.sh * 2; k.x_sw = (int)a->x.sw * 2;
  k.y_sn = (int)b->y.sn * 2; k.y_sh = (int)b->y.sh * 2; k.y_sw = (int)b->y.sw * 2;
  k.r_sn = (int)b->res.sn * 2; k.r_sh = (int)b->res.sh * 2; k.r_sw = (int)b->res.sw * 2;
  k.H = a->x.h; k.W = a->x.w; k.Cout = b->y.c; k.C = C;
  k.kchunks = C / 64; k.w1row_bytes = a->wtaps * C * 2; k.w2row_bytes = b->wtaps * C * 2;
  const int bn = (C == 64 || b->cout_pad % 128) ? 64 : 128;
  const int ntc = b->cout_pad / bn, nmt = C / bn;
  if (nmt > 2 || C % bn) return -1;
  // When the fused launch pays (hipGraph-timed per Bottleneck against the two-launch form, scripts/pair_ubench.py -> profiles/r5d_pair_ubench.txt,
  // us: two launches / fused 4-row tiles / fused 8-row tiles):
  //   C  64 @128x256  20.3 / 13.0 / 15.7     C 128 @64x128  22.2 / 16.0 / 21.8     C 256 @32x64  23.9 / 28.8 / -
  //   C  64 @ 64x128  15.0 / 11.3 / 15.2     C 128 @32x64   14.2 / 16.0 / 21.5     C 256 @16x32  14.3 / 28.4 / -
  //   C  64 @ 32x64    9.1 / 11.1 / 15.0     C 128 @16x32   11.7 / 15.6 / 21.4
  // -> 4-row tiles always (twice the workgroups; the halo recompute is a ninth of the work), only from 96 such tiles on (below that the split-K
  // small-map kernels of the two-launch form fill the chip better), never the two-pass 256-channel form (GEMM 1 twice per N tile).
  // myolo_set_option("pair_th", 4 | 8) forces a tile height and lifts both limits (tests run every variant).
  const int64_t tiles4 = (int64_t)a->x.n * ((k.W + 15) / 16) * ((k.H + 3) / 4);
  int th = 4;
  if (g_pair_th == 4 || g_pair_th == 8) th = (g_pair_th == 8 && nmt == 2) ? 4 : g_pair_th;
  else if (nmt == 2 || tiles4 < 96) return -1;
  k.HW = 18; k.HP = (th + 2) * 18;
  const int hppad = (k.HP + 15) / 16 * 16;
  k.pshift = C == 64 ? 7 : (C == 128 ? 8 : 9);
  k.segmask = C == 64 ? 7 : 15;
  const int ppp = 1024 >> k.pshift;
  k.npieces = (hppad + ppp - 1) / ppp;
  k.halo_bytes = k.npieces * 1024;
  for (int t = 0; t < 9; ++t) {
    k.tap_hoff[t] = (b->tap_dy[t] + 1) * k.HW + (b->tap_dx[t] + 1);
    k.tap_woff[t] = b->tap_w[t] * C * 2;
  }
  k.tiles_x = (k.W + 15) / 16; k.tiles_y = (k.H + th - 1) / th;
  const int64_t nt = (int64_t)a->x.n * k.tiles_x * k.tiles_y;
  if (nt <= 0 || nt > 0x3fffffff) return -1;
  k.ntiles = (int)nt;
  k.tiles_per_xcd = (k.ntiles + 7) / 8;
  hipStream_t st = (hipStream_t)stream;
  if (bn == 64) {
    if (nmt == 1) return th == 8 ? launch<8, 64, 2, 2, 3, 1>(k, 3, ntc, st) : launch<4, 64, 2, 2, 3, 1>(k, 3, ntc, st);
    return launch<4, 64, 2, 2, 3, 2>(k, 3, ntc, st);
  }
  if (nmt == 1) return th == 8 ? launch<8, 128, 4, 2, 4, 1>(k, 1, ntc, st) : launch<4, 128, 4, 2, 4, 1>(k, 1, ntc, st);
  return launch<4, 128, 4, 2, 3, 2>(k, 1, ntc, st);
}

// include/myolo.h: b(a(x)) where a's output has no other reader; a->y is NOT written when the fused kernel runs
extern "C" int myolo_conv_pair(const myolo_conv_desc* a, const myolo_conv_desc* b, void* stream) {
  if (!a || !b || !a->x.ptr || !a->y.ptr || !a->w || !b->x.ptr || !b->y.ptr || !b->w) return MYOLO_EINVAL;
  const int r = pair_try(a, b, stream);
  if (r != -1) return r;
  const int e = myolo_conv(a, stream);
  return e ? e : myolo_conv(b, stream);
}
