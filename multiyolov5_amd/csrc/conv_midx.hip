// conv_mid with the INPUT of a k x k stride-1 layer resident in LDS as a halo tile (round 4 experiment -> kernel): the workgroup owns
// TH x 16 output pixels of one image, stages the (TH + span_y) x (16 + span_x) input pixels ONCE per tile by LDS-DMA (all channels), and
// streams only the weight tiles through the ring -- conv_mid re-fetches the 128-byte pixel rows once per tap (9 x for a 3x3 layer).
//
//   LDS: [halo: HP pixels x PITCH bytes (PITCH = Cin * 2 = 128 / 256 / 512)] [weight ring: NST x BN rows x 128 B] [statistics].
//   A pixel's 16-byte segment s sits at physical segment s ^ (hp & 7) (128-byte pixels) / low four bits ^ (hp & 15) (wider ones): the
//   fragment reads of 16 consecutive halo pixels are conflict-free (128 B) / at most 2-way (256, 512 B) at EVERY alignment of the tap
//   shift (brute-forced over all bases).  The DMA image is lane-linear: lane l of a 1 KB piece fills pixel l / nseg, physical segment
//   l % nseg and fetches the logical segment that belongs there.
//   K loop as conv_mid.hip: counted vmcnt waits, one raw barrier per step, fragment reads + waits in inline asm, the next weight stage's
//   LDS-DMA pieces between the MFMA groups; D^T = W . X^T epilogue with 16-byte NHWC stores and DPP statistics.
//
// Replaces: the k x k stride-1 nn.Conv2d of Bottleneck.cv2 / the head's 3x3 layers in training mode (reference models/common.py:34-46,
// 95-105) and their dgrad (same shape: flipped taps).  Raw epilogue (+ statistics) (+ residual) (+ accumulate).
#include "myolo_dev.h"
#include <string.h>
#include <stdlib.h>

namespace midx {

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

struct MidX {
  const char* x; const char* w; char* y; const char* res; float* stats;
  int x_sn, x_sh, x_sw, y_sn, y_sh, y_sw, r_sn, r_sh, r_sw;      // bytes
  int Hi, Wi, Ho, Wo, Cout;
  int ntaps, kchunks, nsteps, wrow_bytes, accumulate;
  int tiles_x, tiles_y, ntiles, tiles_per_xcd;
  int mindy, mindx, HW, HP, npieces;            // halo geometry: HP = HH * HW pixels, npieces 1 KB LDS-DMA pieces
  int pshift, segmask, halo_bytes;               // PITCH = 1 << pshift; swizzle mask over the low segment bits (7 / 15)
  int tap_hoff[MYOLO_MAX_TAPS];                  // (dy - mindy) * HW + (dx - mindx): halo pixel offset of the tap
  int tap_woff[MYOLO_MAX_TAPS];                  // tap_w[t] * cin_pad * 2 bytes
};

__device__ __forceinline__ float row_sum16(float v) {
#define MIDX_SHR(n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + (n), 0xf, 0xf, true))
  v += MIDX_SHR(1); v += MIDX_SHR(2); v += MIDX_SHR(4); v += MIDX_SHR(8);
#undef MIDX_SHR
  return v;
}

// TH x 16 output pixels x BN channels per tile, WP x WC waves, NST weight stages
// (measured and dropped: the WHOLE weight tile of 3x3 64 -> 64 resident in LDS, K loop without loads, waits or barriers: 24.1 us with 8 waves,
//  31.9 with 4 -- one workgroup per CU -- against 20.4 us for three 47 KB workgroups per CU streaming their weights)
// TW = 16 or 32 output columns per tile row
template <int TH, int BN, int WP, int WC, int NST, int TW = 16>
__global__ __launch_bounds__(64 * WP * WC) void conv_midx_kernel(const MidX p) {
  constexpr int BM = TH * TW, FPR = TW / 16;      // fragments per tile row
  constexpr int NT = 64 * WP * WC, NW = WP * WC;
  constexpr int PW = BM / WP, CW = BN / WC;
  constexpr int PF = PW / 16, CF = CW / 16;       // a wave owns PF tile rows (16 pixels each) x CF channel fragments
  static_assert(CF % 2 == 0 && PW % 16 == 0, "wave tile");
  constexpr int RPI = NT / 8;
  constexpr int WR = BN / RPI;                    // weight pieces per thread and stage
  static_assert(WR >= 1 && BN % RPI == 0, "weight rows over the loader lanes");
  constexpr int WSTAGE = BN * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wp = wave / WC, wc = wave % WC;
  const int lq = lane >> 4, l15 = lane & 15;
  const int tn = blockIdx.y;
  const int xcd = blockIdx.x & 7, bslot = blockIdx.x >> 3, bstride = gridDim.x >> 3;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t*)smem;
  const unsigned ring0 = lds0 + p.halo_bytes;
  char* ring = smem + p.halo_bytes;

  const int lrow = tid >> 3;
  const int lsg = (tid & 7) ^ ((lrow >> 1) & 7);
  int wbase[WR];
#pragma unroll
  for (int j = 0; j < WR; ++j) wbase[j] = (tn * BN + panel_chan(j * RPI + lrow)) * p.wrow_bytes + lsg * 16;
  unsigned foff[2];                               // weight fragment reads (conv_mid.hip layout)
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) foff[kk] = l15 * 128 + (((kk * 4 + lq) ^ ((l15 >> 1) & 7)) << 4);

  float st_s[CF / 2][8], st_q[CF / 2][8];
#pragma unroll
  for (int q = 0; q < CF / 2; ++q)
#pragma unroll
    for (int i = 0; i < 8; ++i) { st_s[q][i] = 0.f; st_q[q][i] = 0.f; }

  const int nseg = 1 << (p.pshift - 4);           // 16-byte segments per pixel
  const int ppp = 1024 >> p.pshift;               // pixels per LDS-DMA piece

  for (int tslot = bslot; tslot < p.tiles_per_xcd; tslot += bstride) {
    const int tile = xcd * p.tiles_per_xcd + tslot;
    if (tile >= p.ntiles) break;
    const int txy = p.tiles_x * p.tiles_y;
    const int n = tile / txy; const int trem = tile - n * txy;
    const int ty0 = (trem / p.tiles_x) * TH, tx0 = (trem - (trem / p.tiles_x) * p.tiles_x) * TW;

    // ---- halo fill: every wave takes pieces wave, wave + NW, ... ----
    for (int pi = wave; pi < p.npieces; pi += NW) {
      const int hp = pi * ppp + (lane >> (p.pshift - 4));
      const int phys = lane & (nseg - 1);
      const int lseg = (phys & ~p.segmask) | ((phys ^ hp) & p.segmask);
      const int hy = hp / p.HW, hx = hp - hy * p.HW;
      const int iy = ty0 + p.mindy + hy, ix = tx0 + p.mindx + hx;
      const bool ok = hp < p.HP && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
      const char* src = ok ? p.x + (unsigned)(n * p.x_sn + iy * p.x_sh + ix * p.x_sw + lseg * 16) : zero_page();
      __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(smem + pi * 1024), 16, 0, 0);
    }

    f4_t acc[CF][PF];
#pragma unroll
    for (int c = 0; c < CF; ++c)
#pragma unroll
      for (int q = 0; q < PF; ++q) acc[c][q] = f4_t{0.f, 0.f, 0.f, 0.f};
    int hbase[PF];                                  // halo pixel of this lane's output pixel at tap offset 0
#pragma unroll
    for (int q = 0; q < PF; ++q) hbase[q] = ((wp * PF + q) / FPR) * p.HW + ((wp * PF + q) % FPR) * 16 + l15;

    int i_tap = 0, i_kc = 0;
    int n_wt = p.tap_woff[0];
    const char* src[WR];
    auto addresses = [&]() {
      const int wo = n_wt + i_kc * 128;
#pragma unroll
      for (int j = 0; j < WR; ++j) src[j] = p.w + (unsigned)(wbase[j] + wo);
      if (++i_kc == p.kchunks) { i_kc = 0; ++i_tap; }
      n_wt = p.tap_woff[i_tap < p.ntaps ? i_tap : 0];
    };
    auto piece = [&](int i, int buf) {
      __builtin_amdgcn_global_load_lds((gptr_t*)src[i], (lptr_t*)(ring + buf * WSTAGE + wave * 1024 + i * RPI * 128), 16, 0, 0);
    };
    int c_tap = 0, c_kc = 0;                        // compute cursor
    int c_hoff = p.tap_hoff[0], n_hoff = p.tap_hoff[p.ntaps > 1 ? 1 : 0];
    constexpr int G = 2 * CF;
    auto step = [&](int buf, int nb, const bool loads) {
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
      if (++c_kc == p.kchunks) { c_kc = 0; ++c_tap; c_hoff = n_hoff; n_hoff = p.tap_hoff[c_tap + 1 < p.ntaps ? c_tap + 1 : 0]; }
      u32x4_t wf[2][CF], xf[2][PF];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int c = 0; c < CF; ++c) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[kk][c]) : "v"(aw + foff[kk]), "n"(c * 2048) : "memory");
#pragma unroll
        for (int q = 0; q < PF; ++q) asm volatile("ds_read_b128 %0, %1" : "=v"(xf[kk][q]) : "v"(ax[kk][q]) : "memory");
      }
      if (loads) addresses();
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

    // weight ring (conv_mid.hip protocol); the halo pieces are older than every weight piece: a counted wait covers them
    const int inflight = p.nsteps < NST - 1 ? p.nsteps : NST - 1;
#pragma unroll
    for (int j = 0; j < NST - 1; ++j)
      if (j < inflight) {
        addresses();
#pragma unroll
        for (int i = 0; i < WR; ++i) piece(i, j);
      }
    int buf = 0;
    const int steady = p.nsteps - (NST - 1);
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
    __builtin_amdgcn_s_barrier();                  // (the next tile's halo fill overwrites what the slowest wave may still be reading)

    // ---- epilogue ----
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int oy = ty0 + (wp * PF + q) / FPR, ox = tx0 + ((wp * PF + q) % FPR) * 16 + l15;
      const bool mvalid = oy < p.Ho && ox < p.Wo;
      const unsigned yoff = (unsigned)(n * p.y_sn + oy * p.y_sh + ox * p.y_sw);
      const unsigned roff = (unsigned)(n * p.r_sn + oy * p.r_sh + ox * p.r_sw);
      uint4 rv[CF / 2], av[CF / 2];
#pragma unroll
      for (int h = 0; h < CF / 2; ++h) {
        const int c0 = tn * BN + wc * CW + 32 * h + 8 * lq;
        const bool ok = mvalid && c0 < p.Cout;
        if (p.res) rv[h] = ldg16(ok ? p.res + roff + c0 * 2 : zero_page());
        if (p.accumulate) av[h] = ldg16(ok ? p.y + yoff + c0 * 2 : zero_page());
      }
#pragma unroll
      for (int h = 0; h < CF / 2; ++h) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = acc[2 * h][q][r]; v[4 + r] = acc[2 * h + 1][q][r]; }
        if (!mvalid) {                              // a ragged tile's outside pixels still see inside taps: keep them out of the statistics
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { st_s[h][i] += v[i]; st_q[h][i] += v[i] * v[i]; }
        const int c0 = tn * BN + wc * CW + 32 * h + 8 * lq;
        if (mvalid && c0 < p.Cout) {
          if (p.res) add_h8(v, u32x4_t{rv[h].x, rv[h].y, rv[h].z, rv[h].w});
          char* yp = p.y + yoff + c0 * 2;
          if (p.accumulate) add_h8(v, u32x4_t{av[h].x, av[h].y, av[h].z, av[h].w});
          const u32x4_t o = pack_h8(v);
          stg16(yp, uint4{o.x, o.y, o.z, o.w});
        }
      }
    }
  }

  if (p.stats != nullptr) {
    float* red = reinterpret_cast<float*>(smem);       // [WP][2][BN]
#pragma unroll
    for (int h = 0; h < CF / 2; ++h)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float s = row_sum16(st_s[h][i]), q = row_sum16(st_q[h][i]);
        if (l15 == 15) {
          const int cl = wc * CW + 32 * h + 8 * lq + i;
          red[(wp * 2) * BN + cl] = s; red[(wp * 2 + 1) * BN + cl] = q;
        }
      }
    __syncthreads();
    for (int t = tid; t < 2 * BN; t += NT) {
      const int which = t / BN, cl = t - which * BN;
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < WP; ++k) a += red[(k * 2 + which) * BN + cl];
      const int c = tn * BN + cl;
      if (c < p.Cout) atomicAdd(p.stats + (blockIdx.x % MYOLO_STAT_COPIES) * 2 * p.Cout + which * p.Cout + c, a);
    }
  }
}

template <int TH, int BN, int WP, int WC, int NST, int TW = 16>
int launch(const MidX& k, int per_cu, int ntile_c, hipStream_t st) {
  constexpr int NT = 64 * WP * WC;
  const int smem = k.halo_bytes + NST * BN * 128;
  if (smem > 160 * 1024) return -1;
  int fit = (160 * 1024) / (smem + 256);
  if (fit < per_cu) per_cu = fit < 1 ? 1 : fit;
  int per_xcd = (256 * per_cu / ntile_c + 7) / 8;
  if (per_xcd < 1) per_xcd = 1;
  if (per_xcd > k.tiles_per_xcd) per_xcd = k.tiles_per_xcd;
  auto kern = conv_midx_kernel<TH, BN, WP, WC, NST, TW>;
  MYOLO_ENSURE_DYN_SMEM(kern, smem);
  hipLaunchKernelGGL(kern, dim3(per_xcd * 8, ntile_c), dim3(NT), smem, st, k);
  MYOLO_CHECK_LAUNCH();
  return 0;
}

}  // namespace midx

static int g_midx_mode = -1;       // 0 off, 1 on (MYOLO_CONV_MIDX)
static int g_midx_var = 0;
int myolo_conv_midx_set(const char* name, int value) {
  if (!strcmp(name, "midx_mode")) { g_midx_mode = value; return 0; }
  if (!strcmp(name, "midx_var")) { g_midx_var = value; return 0; }
  return MYOLO_EINVAL;
}

// -1: the layer does not qualify.  No bnb fold: the caller runs the reduce pass.
int myolo_conv_midx_try(const myolo_conv_desc* d, void* stream) {
  using namespace midx;
  if (g_midx_mode < 0) g_midx_mode = getenv("MYOLO_CONV_MIDX") ? atoi(getenv("MYOLO_CONV_MIDX")) : 1;
  if (!g_midx_mode) return -1;
  if (d->x.dtype != MYOLO_F16 || d->det_no > 0 || d->up_shift != 0 || d->scale || d->shift || d->act != MYOLO_ACT_NONE) return -1;
  if (d->stride != 1 || d->ntaps < 2 || d->ntaps > 25) return -1;
  if (d->cin_pad != 64 && d->cin_pad != 128 && d->cin_pad != 256) return -1;
  if (d->x.c != d->cin_pad || d->cout_pad % 64 || d->y.c % 8) return -1;
  if (d->x.h != d->y.h || d->x.w != d->y.w || d->x.n != d->y.n) return -1;
  if (d->res.ptr && d->res.c < d->y.c) return -1;
  auto extent = [](const myolo_tensor& t) { return ((int64_t)t.n * t.sn + (int64_t)t.h * t.sh + (int64_t)t.w * t.sw + t.c) * 2; };
  if (extent(d->x) >= (1ll << 31) || extent(d->y) >= (1ll << 31) || (d->res.ptr && extent(d->res) >= (1ll << 31))) return -1;
  if ((int64_t)d->cout_pad * d->wtaps * d->cin_pad * 2 >= (1ll << 31)) return -1;
  int mindy = 1 << 20, maxdy = -(1 << 20), mindx = 1 << 20, maxdx = -(1 << 20);
  for (int t = 0; t < d->ntaps; ++t) {
    mindy = d->tap_dy[t] < mindy ? d->tap_dy[t] : mindy; maxdy = d->tap_dy[t] > maxdy ? d->tap_dy[t] : maxdy;
    mindx = d->tap_dx[t] < mindx ? d->tap_dx[t] : mindx; maxdx = d->tap_dx[t] > maxdx ? d->tap_dx[t] : maxdx;
  }
  if (maxdy - mindy > 8 || maxdx - mindx > 8) return -1;
  const int64_t M = (int64_t)d->y.n * d->y.h * d->y.w;
  if (M < 4096) return -1;
  MidX k;
  k.x = (const char*)d->x.ptr; k.w = (const char*)d->w; k.y = (char*)d->y.ptr; k.res = (const char*)d->res.ptr; k.stats = d->stats;
  k.x_sn = (int)d->x.sn * 2; k.x_sh = (int)d->x.sh * 2; k.x_sw = (int)d->x.sw * 2;
  k.y_sn = (int)d->y.sn * 2; k.y_sh = (int)d->y.sh * 2; k.y_sw = (int)d->y.sw * 2;
  k.r_sn = (int)d->res.sn * 2; k.r_sh = (int)d->res.sh * 2; k.r_sw = (int)d->res.sw * 2;
  k.Hi = d->x.h; k.Wi = d->x.w; k.Ho = d->y.h; k.Wo = d->y.w; k.Cout = d->y.c;
  k.ntaps = d->ntaps; k.kchunks = d->cin_pad / 64; k.nsteps = k.ntaps * k.kchunks; k.wrow_bytes = d->wtaps * d->cin_pad * 2;
  k.accumulate = d->accumulate;
  const int bn = d->cout_pad % 128 == 0 ? 128 : 64;
  const bool want_bnb = d->bnb && d->nbnb > 0;
  int var = g_midx_var;
  if (!var) {
    // hipGraph-timed at batch 16 against conv_mid (scripts/conv_train_ubench.py PROBE=midx, us with statistics): 3x3 64->64 @64x128 31.6 -> 20.5
    // (dilation 2 / 3: 28.7 -> 21.5, 29.0 -> 22.1), 128->128 @32x64 18.3 -> 17.3, head 256->128 @64x128 100.2 -> 94.0 and its dgrad 101.9 -> 88.7
    // with 16 x 16 pixel tiles; 256->256 @16x32 24.8 -> 26.3 (stays on conv_mid).  This kernel has no bnb fold: a layer that carries one would
    // pay a reduce launch for a 1 us gain -- only the 64-channel layers (whose maps are too large for the fold anyway) take it then.
    if (bn == 64) var = 2;
    else if (M >= 65536 && d->ntaps * (d->cin_pad / 64) >= 16) var = 3;
    else if (d->cin_pad == 128) var = 1;
    else return -1;
    if (want_bnb && d->cin_pad != 64) return -1;
  }
  if (bn == 64 || var > 3) var = 2;
  const int TH = var == 3 ? 16 : 8, TW = 16;
  k.mindy = mindy; k.mindx = mindx;
  const int HH = TH + (maxdy - mindy);
  k.HW = TW + (maxdx - mindx);
  k.HP = HH * k.HW;
  k.pshift = d->cin_pad == 64 ? 7 : (d->cin_pad == 128 ? 8 : 9);
  k.segmask = d->cin_pad == 64 ? 7 : 15;
  const int ppp = 1024 >> k.pshift;
  k.npieces = (k.HP + ppp - 1) / ppp;
  k.halo_bytes = k.npieces * 1024;
  for (int t = 0; t < MYOLO_MAX_TAPS; ++t) {
    const bool in = t < d->ntaps;
    k.tap_hoff[t] = in ? (d->tap_dy[t] - mindy) * k.HW + (d->tap_dx[t] - mindx) : 0;
    k.tap_woff[t] = in ? d->tap_w[t] * d->cin_pad * 2 : 0;
  }
  k.tiles_x = (k.Wo + TW - 1) / TW; k.tiles_y = (k.Ho + TH - 1) / TH;
  k.ntiles = d->y.n * k.tiles_x * k.tiles_y;
  k.tiles_per_xcd = (k.ntiles + 7) / 8;
  hipStream_t st = (hipStream_t)stream;
  const int ntc = d->cout_pad / (var == 2 ? 64 : 128);
  // (8 x 32 pixel tiles with 8 waves and a two-stage ring with four workgroups per CU measured the same as var 2: 21.5 / 22.0 vs 20.6 us)
  if (var == 1) return launch<8, 128, 4, 2, 4>(k, 1, ntc, st);
  if (var == 3) return launch<16, 128, 4, 2, 3>(k, 1, ntc, st);
  return launch<8, 64, 2, 2, 3>(k, 3, ntc, st);
}
