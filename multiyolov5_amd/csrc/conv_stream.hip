// Streaming implicit-GEMM convolution for the HBM-bound layers (fp16): small K = taps*Cin, huge M = pixels.
//
//   The per-layer roofline puts 69 of the 79 convolutions of yolov5s+PSP on the memory side: the job is to
//   stream activations at HBM rate, the MFMA work hides underneath.  So, unlike the LDS-tiled kernel in conv_igemm.hip
//   (one 64-byte K chunk per workgroup barrier), here
//     * the whole weight panel of the workgroup's N tile ([BN][taps*cin_pad], <= ~96 KB) is staged in LDS ONCE;
//     * every wave owns 32-pixel row tiles and loads its MFMA A fragments straight from global memory (lane = pixel row
//       lane&15, 16-byte K segment lane>>4: whole 64-byte pieces of NHWC rows), through a 4-deep register ring: three
//       chunks (12 KB per wave) are always in flight, and there is NO barrier in the main loop -- waves never wait for
//       each other, only for their own loads (counted vmcnt);
//     * the epilogue (BN statistics, scale/shift, SiLU, residual) goes through a small wave-private LDS transpose so that
//       NHWC stores are 16-byte row-contiguous.
//   Workgroups are persistent over XCD-contiguous ranges of row tiles (3x3 halos of neighbouring tiles hit the same L2).
//
// Same contract as myolo_conv (include/myolo.h); selected by myolo_conv when the layer qualifies (conv_igemm.hip).
#include "myolo_dev.h"
#include <stdlib.h>
#include <string.h>

namespace stream {

constexpr int WAVES = 8;
constexpr int THREADS = WAVES * 64;
constexpr int RING = 4;

struct ConvS {
  const char* x; int64_t x_sn, x_sh, x_sw; int Hi, Wi, Cin;
  char* y; int64_t y_sn, y_sh, y_sw; int Ho, Wo, Cout, N;
  const char* w; int cin_pad, cout_pad, wtaps, ntaps, stride, up;
  int tap_dy[MYOLO_MAX_TAPS], tap_dx[MYOLO_MAX_TAPS], tap_w[MYOLO_MAX_TAPS];
  const float* scale; const float* shift; int act; int accumulate;
  const char* res; int64_t r_sn, r_sh, r_sw;
  float* stats; int M; int ntiles; int tiles_per_xcd; int pitchB; int xdense, ydense; int dbg;
  int x_bytes, y_bytes, r_bytes;     // byte spans of the views (buffer descriptors)
  BnbArgs bnb;                       // BatchNorm-backward statistics folded into the epilogue (BNS); segment tensors are pixel-dense
};

__device__ __forceinline__ int swz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }   // H = {0,2,3,1}

// EPI 0: raw output (+ BatchNorm statistics) -- the training forward and every dgrad;  EPI 1: scale/shift + activation (eval)
// BNS = 1 (with EPI 0): the stored gradient completes gout of a BatchNorm layer -> its backward sums (myolo_conv_desc.bnb)
template <int BN, int KB, int EPI, int EXTRA, int BNS = 0>
__global__ __launch_bounds__(THREADS) void conv_stream_kernel(const ConvS p) {
  constexpr int NF = BN / 16;
  constexpr int KCH = 32 * KB;                 // halves per A chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sB = smem;                             // [BN][pitchB] weight panel
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int tn = blockIdx.y;

  // (the weight panel is staged further down, BEHIND the first activation loads: both are ~2 us round trips at kernel start)
  // epilogue constants live in LDS: a global load inside the epilogue would sit BEHIND the prefetched activations in the
  // in-order vmcnt queue and drain the whole pipeline every tile
  float* sT = reinterpret_cast<float*>(smem + (size_t)BN * p.pitchB);     // [4][BN]: scale, shift (EPI 1) | mean, invstd, sc, sh (BNS)
  if (EPI == 1)
    for (int c = tid; c < BN; c += THREADS) {
      const int cg = tn * BN + c;
      sT[c] = (p.scale && cg < p.Cout) ? p.scale[cg] : 1.0f;
      sT[BN + c] = (p.shift && cg < p.Cout) ? p.shift[cg] : 0.0f;
    }
  int sgi = -1;
  if (BNS) {
    for (int i = 0; i < p.bnb.n; ++i)
      if (tn * BN >= p.bnb.seg[i].c0 && tn * BN < p.bnb.seg[i].c1) sgi = i;
    if (sgi >= 0) {
      const BnbSeg& sg = p.bnb.seg[sgi];
      const int Cs = sg.c1 - sg.c0;
      for (int c = tid; c < BN; c += THREADS) {
        const int ci = tn * BN + c - sg.c0;
        const bool in = ci < Cs;
        const float mean = in ? sg.saved[ci] : 0.f, istd = in ? sg.saved[Cs + ci] : 0.f;
        const float sc = in ? sg.gamma[ci] * istd : 0.f;
        sT[c] = mean; sT[BN + c] = istd; sT[2 * BN + c] = sc; sT[3 * BN + c] = in ? sg.beta[ci] - mean * sc : 0.f;
      }
    }
  }
  int* sTap = reinterpret_cast<int*>(sT + 4 * BN);                        // [2][MAX_TAPS] tap_dy, tap_dx
  for (int t = tid; t < p.ntaps; t += THREADS) { sTap[t] = p.tap_dy[t]; sTap[MYOLO_MAX_TAPS + t] = p.tap_dx[t]; }
  __syncthreads();

  const int kchunks = p.cin_pad / KCH;
  const int cpt = p.ntaps * kchunks;            // chunks per tile
  const int HWo = p.Ho * p.Wo;
  const int Hlog = p.Hi << p.up, Wlog = p.Wi << p.up;

  // tiles of this wave: XCD-contiguous ranges, interleaved over the waves resident on the XCD
  const int xcd = blockIdx.x & 7;
  const int wslot = (blockIdx.x >> 3) * WAVES + wave;
  const int wstride = (gridDim.x >> 3) * WAVES;
  const int tile_lo = xcd * p.tiles_per_xcd;
  int tile_hi = tile_lo + p.tiles_per_xcd;
  if (tile_hi > p.ntiles) tile_hi = p.ntiles;
  const int my_first = tile_lo + wslot;
  const int ntl = my_first < tile_hi ? (tile_hi - my_first + wstride - 1) / wstride : 0;   // tiles of this wave
  const int nchunks = ntl * cpt;

  // The MFMA is issued as D^T = W . X^T: acc[nf][r] belongs to weight-panel row nf*16 + 4*lq + r and pixel (lane&15).  The panel rows
  // are PERMUTED (panel_chan): rows 4*lq + r of fragments 2q and 2q+1 hold output channels 32q + 8*lq + {r, 4 + r}, so that every lane
  // owns 8 CONSECUTIVE channels of one pixel: 16-byte NHWC stores straight from registers (no LDS transpose), 64 contiguous bytes per
  // pixel and store instruction (with the natural row order a lane held 4 channels: 8-byte stores, 32-byte pieces -- the 64 -> 64 1x1
  // layer at 16x128x256 took 44.6 us against 35.8 us with 16-byte stores, r3 store experiment).
  float st_s[NF][4], st_q[NF][4];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int r = 0; r < 4; ++r) { st_s[nf][r] = 0.f; st_q[nf][r] = 0.f; }

  // ---- buffer descriptors: hardware bounds checking does the masking -- an out-of-range voffset loads zeros and drops stores, so
  // neither loads nor stores need branches or a zero page, and addressing is one 32-bit add per access (wave-uniform base) ----
  constexpr int OOB = 0x7fff0000;
  const bool half_tail = (p.Cout & 4) != 0;      // wave-uniform: the last 8-channel group of the layer is half a group
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.res ? p.res : p.y), 0, p.res ? p.r_bytes : 0, 0x00020000);

  // ---- issue cursor ----
  int i_tile = my_first, i_tap = 0, i_kc = 0, i_left = nchunks;
  int i_off[2];                        // byte offset of the tile row's centre pixel (or OOB)
  int i_y0[2], i_x0[2];
  auto decode_issue = [&]() {
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      const int m = i_tile * 32 + mf * 16 + l15;
      const bool ok = m < p.M;
      const int mm = ok ? m : 0;
      if (p.xdense) {                       // 1x1 stride 1 over a pixel-dense view: pixel m is at base + m*sw
        i_off[mf] = ok ? mm * (int)p.x_sw * 2 : OOB;
        i_y0[mf] = 0; i_x0[mf] = 0;
      } else {
        const int n = mm / HWo; const int rem = mm - n * HWo; const int oy = rem / p.Wo; const int ox = rem - oy * p.Wo;
        i_off[mf] = ok ? n * (int)p.x_sn * 2 : OOB;
        i_y0[mf] = oy * p.stride; i_x0[mf] = ox * p.stride;
      }
    }
  };
  if (i_left > 0) decode_issue();
  uint4 ring[RING][2 * KB];
  auto bload = [&](int off) -> uint4 {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
    return uint4{v.x, v.y, v.z, v.w};
  };
  auto issue = [&](uint4* dst) {
    // branch-free: every load is issued unconditionally; dead lanes / padded channels / finished waves use an out-of-range offset
    const bool live = i_left > 0;
    const bool ld = live && !(p.dbg & 2);
    if (p.xdense) {
#pragma unroll
      for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          const int c0 = i_kc * KCH + kb * 32 + lq * 8;
          dst[mf * KB + kb] = bload((ld && c0 < p.Cin) ? i_off[mf] + c0 * 2 : OOB);
        }
    } else {
      // wave-uniform tap index -> LDS broadcast read of the tap table (a VGPR-indexed kernarg read is a global_load that would
      // queue BEHIND the prefetched activations in the in-order vmcnt queue)
      const int ut = __builtin_amdgcn_readfirstlane(i_tap);
      const int dy = sTap[ut], dx = sTap[MYOLO_MAX_TAPS + ut];
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) {
        int iy = i_y0[mf] + dy, ix = i_x0[mf] + dx;
        const bool ok = ld && (unsigned)iy < (unsigned)Hlog && (unsigned)ix < (unsigned)Wlog;
        iy >>= p.up; ix >>= p.up;
        const int ro = i_off[mf] + (iy * (int)p.x_sh + ix * (int)p.x_sw) * 2;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          const int c0 = i_kc * KCH + kb * 32 + lq * 8;
          dst[mf * KB + kb] = bload((ok && c0 < p.Cin) ? ro + c0 * 2 : OOB);
        }
      }
    }
    if (live) {
      --i_left;
      if (++i_kc == kchunks) {
        i_kc = 0;
        if (++i_tap == p.ntaps) { i_tap = 0; i_tile += wstride; if (i_left > 0) decode_issue(); }
      }
    }
  };

  // ---- compute cursor ----
  int c_tile = my_first, c_tap = 0, c_kc = 0;
  f4_t acc[2][NF];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

  auto epilogue = [&](int tile) {
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      const int m = tile * 32 + mf * 16 + l15;
      const bool mok = m < p.M && !(p.dbg & 1);
      int yoff, roff = 0;                      // byte offsets (buffer addressing; out-of-range -> the store is dropped)
      if (p.ydense) {
        yoff = m * (int)p.y_sw * 2;
        if (EXTRA) roff = m * (int)p.r_sw * 2;
      } else {
        const int mm = mok ? m : 0;
        const int n = mm / HWo; const int rem = mm - n * HWo; const int oy = rem / p.Wo; const int ox = rem - oy * p.Wo;
        yoff = (n * (int)p.y_sn + oy * (int)p.y_sh + ox * (int)p.y_sw) * 2;
        if (EXTRA) roff = (n * (int)p.r_sn + oy * (int)p.r_sh + ox * (int)p.r_sw) * 2;
      }
#pragma unroll
      for (int q = 0; q < NF / 2; ++q) {
        // weight-panel row permutation (panel_chan, myolo_dev.h): fragments 2q, 2q+1 hold channels cl .. cl+7 of this lane's pixel
        const int cl = q * 32 + 8 * lq;
        const int c0 = tn * BN + cl;
        float v[8];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v0 = acc[mf][2 * q + h][r];
            acc[mf][2 * q + h][r] = 0.f;
            if (EPI == 0) { if (!BNS) { st_s[2 * q + h][r] += v0; st_q[2 * q + h][r] += v0 * v0; } v[4 * h + r] = v0; }
            else v[4 * h + r] = act_f(v0 * sT[cl + 4 * h + r] + sT[BN + cl + 4 * h + r], p.act);
          }
        // Cout % 4 == 0 (checked on the host): a lane's 8 channels are whole, or (last group of a Cout % 8 == 4 layer) their first half
        const bool ok8 = mok && c0 + 8 <= p.Cout, ok4 = mok && !ok8 && c0 + 4 <= p.Cout;
        if (EXTRA) {                           // residual / accumulate: loads inside the epilogue (they drain the prefetch queue)
          if (p.res) add_h8(v, buf_load_h8(rr, roff + c0 * 2, ok8, ok4, half_tail));
          if (p.accumulate) add_h8(v, buf_load_h8(ry, yoff + c0 * 2, ok8, ok4, half_tail));
        }
        const u32x4_t o = pack_h8(v);
        __builtin_amdgcn_raw_buffer_store_b128(o, ry, ok8 ? yoff + c0 * 2 : OOB, 0, 0);
        if (half_tail) __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{o.x, o.y}, ry, ok4 ? yoff + c0 * 2 : OOB, 0, 0);
        if (BNS) {
          if (sgi >= 0) {                // (wave-uniform) dz = gout * act'(z) of the normalised layer, from its raw output at pixel m
            const BnbSeg& sg = p.bnb.seg[sgi];
            const bool okb = ok8 && c0 < sg.c1;                    // segments start and end on N-tile boundaries (bnb_aligned)
            const char* yp = okb ? sg.y + ((int64_t)m * sg.y_sw + (c0 - sg.c0)) * 2 : zero_page();
            const uint4 yr = ldg16(yp);
            const half_t* yh = reinterpret_cast<const half_t*>(&yr);
            const half_t* oh = reinterpret_cast<const half_t*>(&o);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int e = 4 * h + r;
                const float yv = (float)yh[e], g = okb ? (float)oh[e] : 0.f;
                const float dz = g * act_grad_f(fmaf(yv, sT[2 * BN + cl + e], sT[3 * BN + cl + e]), sg.act);
                st_s[2 * q + h][r] += dz;
                st_q[2 * q + h][r] += dz * (yv - sT[cl + e]) * sT[BN + cl + e];
              }
          }
        }
      }
    }
  };

  auto compute = [&](const uint4* a, const bool live) {
    // panel rows are 64-byte blocks with the 16-byte segment XOR-swizzled by H[(row>>2)&3] (pitch/64 is odd): conflict-free
    // for the ds_read_b128 lane groups of the fragment pattern (row = lane&15, segment = lane>>4)
    const char* brow = sB + l15 * p.pitchB + (c_tap * p.cin_pad + c_kc * KCH) * 2 + ((lq ^ swz(l15)) << 4);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const uint4 b = *reinterpret_cast<const uint4*>(brow + nf * 16 * p.pitchB + kb * 64);
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)      // weights as the A operand, pixels as the B operand: D[cout][pixel]
          acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const h8_t*>(&b),
                                                               *reinterpret_cast<const h8_t*>(&a[mf * KB + kb]), acc[mf][nf], 0, 0, 0);
      }
    }
    if (live && ++c_kc == kchunks) {
      c_kc = 0;
      if (++c_tap == p.ntaps) { c_tap = 0; epilogue(c_tile); c_tile += wstride; }
    }
  };

  // ---- software pipeline: RING-1 chunks in flight ----
#pragma unroll
  for (int j = 0; j < RING - 1; ++j) issue(ring[j]);
  // ---- stage the weight panel once (the activation loads above are already in flight) ----
  stage_weight_panel<BN, THREADS, true>(sB, p.w, tn, p.pitchB, p.cin_pad, p.ntaps, p.wtaps, p.tap_w, tid);
  __syncthreads();
  for (int q = 0; q < nchunks; q += RING) {
#pragma unroll
    for (int j = 0; j < RING; ++j) {
      issue(ring[(j + RING - 1) % RING]);
      compute(ring[j], q + j < nchunks);          // dead chunks (zero page) only add zeros
    }
  }

  if (EPI == 0 && (BNS ? sgi >= 0 : p.stats != nullptr)) {
    // per-channel sums: lanes -> wave (shuffles over the 16 pixel lanes) -> workgroup (LDS) -> ONE coalesced atomic per
    // channel per workgroup.  (Per-wave atomics on the same 2*Cout addresses cost ~400 us per launch.)
    __syncthreads();                                   // every wave is done with the weight panel
    float* red = reinterpret_cast<float*>(smem);       // [WAVES][2*BN]
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = st_s[nf][r], q2 = st_q[nf][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o, 64); q2 += __shfl_xor(q2, o, 64); }
        if (l15 == 0) {
          const int cl = panel_chan(nf * 16 + 4 * lq + r);
          red[wave * 2 * BN + cl] = s;
          red[wave * 2 * BN + BN + cl] = q2;
        }
      }
    __syncthreads();
    for (int t = tid; t < 2 * BN; t += THREADS) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) a += red[w * 2 * BN + t];
      const int cl = t < BN ? t : t - BN;
      const int c = tn * BN + cl;
      if (BNS) {
        const BnbSeg& sg = p.bnb.seg[sgi];
        const int Cs = sg.c1 - sg.c0, ci = c - sg.c0;
        if (ci < Cs) atomicAdd(sg.dsum + (blockIdx.x % MYOLO_STAT_COPIES) * 2 * Cs + (t < BN ? ci : Cs + ci), a);
      } else if (c < p.Cout) {
        atomicAdd(p.stats + (blockIdx.x % MYOLO_STAT_COPIES) * 2 * p.Cout + (t < BN ? c : p.Cout + c), a);
      }
    }
  }
}

template <int BN, int KB, int EPI, int EXTRA, int BNS = 0>
int launch3(const ConvS& k, int grid_x, int ntile_n, int smem, hipStream_t st) {
  auto kern = conv_stream_kernel<BN, KB, EPI, EXTRA, BNS>;
  MYOLO_ENSURE_DYN_SMEM(kern, smem);
  hipLaunchKernelGGL(kern, dim3(grid_x, ntile_n), dim3(THREADS), smem, st, k);
  MYOLO_CHECK_LAUNCH();
  return 0;
}
template <int BN, int KB, int EPI>
int launch2(const ConvS& k, int grid_x, int ntile_n, int smem, hipStream_t st) {
  const bool extra = k.res != nullptr || k.accumulate;
  if (EPI == 0 && k.bnb.n > 0)
    return extra ? launch3<BN, KB, 0, 1, 1>(k, grid_x, ntile_n, smem, st) : launch3<BN, KB, 0, 0, 1>(k, grid_x, ntile_n, smem, st);
  return extra ? launch3<BN, KB, EPI, 1>(k, grid_x, ntile_n, smem, st) : launch3<BN, KB, EPI, 0>(k, grid_x, ntile_n, smem, st);
}
template <int BN, int KB>
int launch(const ConvS& k, int grid_x, int ntile_n, int smem, hipStream_t st) {
  const bool raw = !k.scale && !k.shift && k.act == MYOLO_ACT_NONE;
  return raw ? launch2<BN, KB, 0>(k, grid_x, ntile_n, smem, st) : launch2<BN, KB, 1>(k, grid_x, ntile_n, smem, st);
}

}  // namespace stream

static inline int panel_pitch(int K) {                    // bytes; multiple of 64 with an odd number of 64-byte blocks
  int blocks = (K * 2 + 63) / 64;
  if (!(blocks & 1)) ++blocks;
  return blocks * 64;
}

// returns -1 when the layer does not qualify (caller falls back to the LDS-tiled kernel), else a hipError_t / 0
extern int g_nms_dbg;                    // nms.hip
static int g_stream_min_tiles = -1;      // -1: from the environment (MYOLO_STREAM_MIN_TILES) or 2048
static int g_stream_off = -1;
static int g_stream_dbg = -1;           // profiling only: 1 no stores, 2 no activation loads
static int g_stream_per_cu = 0;         // 0: default (2 workgroups per CU when LDS allows)

extern "C" int myolo_set_option(const char* name, int value) {
  if (!name) return MYOLO_EINVAL;
  if (!strcmp(name, "stream_min_tiles")) { g_stream_min_tiles = value; return 0; }
  if (!strcmp(name, "stream_off")) { g_stream_off = value; return 0; }
  if (!strcmp(name, "stream_dbg")) { g_stream_dbg = value; return 0; }
  if (!strcmp(name, "stream_per_cu")) { g_stream_per_cu = value; return 0; }
  if (!strcmp(name, "nms_dbg")) { g_nms_dbg = value; return 0; }
  if (!strncmp(name, "bn_", 3)) return myolo_bn_set(name, value);
  if (!strncmp(name, "spp_", 4) || !strncmp(name, "pool_", 5)) return myolo_pool_set(name, value);
  if (!strncmp(name, "stem_", 5)) return myolo_stem_set(name, value);
  if (!strncmp(name, "igemm_", 6)) return myolo_conv_igemm_set(name, value);
  if (!strncmp(name, "midx_", 5)) return myolo_conv_midx_set(name, value);
  if (!strncmp(name, "pair_", 5)) return myolo_conv_pair_set(name, value);
  if (!strncmp(name, "mid_", 4)) return myolo_conv_mid_set(name, value);
  return myolo_conv_halo_set(name, value);      // "halo_off", "halo_min_tiles"

}

int myolo_conv_stream_try(const myolo_conv_desc* d, void* stream, int* bnb_done) {
  using namespace stream;
  *bnb_done = 0;
  if (g_stream_min_tiles < 0) g_stream_min_tiles = 2048;   // (myolo_set_option("stream_min_tiles" / "stream_off" / "stream_dbg"): tests and sweeps)
  if (g_stream_off < 0) g_stream_off = 0;
  const int min_tiles = g_stream_min_tiles;
  const bool off = g_stream_off != 0;
  if (off || d->x.dtype != MYOLO_F16 || d->det_no > 0 || (d->y.c & 3)) return -1;
  if (d->cin_pad % 32) return -1;
  // N tile: the weight panel [BN][ntaps*cin_pad] (+16 B row padding) must fit beside the 8 wave-private staging areas
  const int K = d->ntaps * d->cin_pad;
  int bn = 0;
  // BN = 128 needs > 200 VGPRs with the per-channel statistics: two N tiles instead.  96 (yolov5m: 96 / 192 / 384 / 768 channels)
  // where it means fewer N tiles than 64: every N tile re-streams the activations
  constexpr int no96 = 0;
  const int cands[3] = {96, 64, 32};
  for (int i = 0; i < 3; ++i) {
    const int b = cands[i];
    if (d->cout_pad % b) continue;
    if (b == 96 && (no96 || (d->cout_pad % 64 == 0 && d->cout_pad / 64 <= d->cout_pad / 96))) continue;
    if (b * panel_pitch(K) + 4 * b * 4 + 256 <= 144 * 1024) { bn = b; break; }
  }
  if (!bn) return -1;
  if (d->ntaps > 1 && d->cout_pad / bn > 2) return -1;  // multi-tap A tiles re-streamed per N tile (the 128 -> 256 3x3 dgrad of the PSP head: 8 N
                                                       // tiles of 32, 270 us here against 122 us for the same shape tiled)
  if (d->cout_pad / bn > 8) return -1;                 // A would be re-streamed too often: the tiled kernel wins (6 N tiles of the
                                                       // 64 -> 384 dgrads still stream: 116 us tiled in the r2 step profile)
  ConvS k;
  k.x = (const char*)d->x.ptr; k.x_sn = d->x.sn; k.x_sh = d->x.sh; k.x_sw = d->x.sw;
  k.Hi = d->x.h; k.Wi = d->x.w; k.Cin = d->x.c;
  k.y = (char*)d->y.ptr; k.y_sn = d->y.sn; k.y_sh = d->y.sh; k.y_sw = d->y.sw;
  k.Ho = d->y.h; k.Wo = d->y.w; k.Cout = d->y.c; k.N = d->y.n;
  k.w = (const char*)d->w; k.cin_pad = d->cin_pad; k.cout_pad = d->cout_pad; k.wtaps = d->wtaps;
  k.ntaps = d->ntaps; k.stride = d->stride; k.up = d->up_shift;
  for (int i = 0; i < MYOLO_MAX_TAPS; ++i) { k.tap_dy[i] = d->tap_dy[i]; k.tap_dx[i] = d->tap_dx[i]; k.tap_w[i] = d->tap_w[i]; }
  k.scale = d->scale; k.shift = d->shift; k.act = d->act; k.accumulate = d->accumulate;
  k.res = (const char*)d->res.ptr; k.r_sn = d->res.sn; k.r_sh = d->res.sh; k.r_sw = d->res.sw;
  k.stats = d->stats;
  k.bnb.n = 0;
  if (d->bnb && d->nbnb > 0 && !d->stats && !d->scale && !d->shift && d->act == MYOLO_ACT_NONE && bnb_aligned(d, bn)) {
    bool dense_ok = true;                      // the epilogue addresses the segment tensors by linear pixel index
    for (int i = 0; i < d->nbnb; ++i) {
      const myolo_tensor& t = d->bnb[i].y;
      dense_ok = dense_ok && t.sh == (int64_t)t.w * t.sw && t.sn == (int64_t)t.h * t.sh && t.n == d->y.n && t.h == d->y.h && t.w == d->y.w;
    }
    if (dense_ok) { bnb_fill(&k.bnb, d); *bnb_done = 1; }
  }
  const int64_t M = (int64_t)k.N * k.Ho * k.Wo;
  if (M <= 0 || M > 0x7fffffff) return MYOLO_EINVAL;
  k.M = (int)M;
  k.ntiles = (int)((M + 31) / 32);
  if (k.ntiles < min_tiles) return -1;                      // small maps: launch/prologue bound, keep the tiled kernel
  k.tiles_per_xcd = (k.ntiles + 7) / 8;
  k.pitchB = panel_pitch(K);
  auto dense = [](int64_t sn, int64_t sh, int64_t sw, int H, int W) { return sh == (int64_t)W * sw && sn == (int64_t)H * sh; };
  k.ydense = dense(k.y_sn, k.y_sh, k.y_sw, k.Ho, k.Wo) && (!k.res || dense(k.r_sn, k.r_sh, k.r_sw, k.Ho, k.Wo));
  k.xdense = d->ntaps == 1 && d->stride == 1 && d->up_shift == 0 && d->tap_dy[0] == 0 && d->tap_dx[0] == 0 &&
             k.Hi == k.Ho && k.Wi == k.Wo && dense(k.x_sn, k.x_sh, k.x_sw, k.Hi, k.Wi);
  if (g_stream_dbg < 0) g_stream_dbg = 0;
  k.dbg = g_stream_dbg;
  if (k.stats && (k.scale || k.shift || k.act != MYOLO_ACT_NONE)) return -1;   // statistics only with the raw epilogue
  auto span = [](const myolo_tensor& t) -> int64_t {
    return (((int64_t)t.n - 1) * t.sn + ((int64_t)t.h - 1) * t.sh + ((int64_t)t.w - 1) * t.sw + t.c) * 2;
  };
  const int64_t xb = span(d->x), yb = span(d->y), rb = d->res.ptr ? span(d->res) : 0;
  if (xb >= 0x7ffe0000LL || yb >= 0x7ffe0000LL || rb >= 0x7ffe0000LL) return -1;      // 32-bit buffer offsets
  k.x_bytes = (int)xb; k.y_bytes = (int)yb; k.r_bytes = (int)rb;
  int smem = bn * k.pitchB + 4 * bn * 4 + 2 * MYOLO_MAX_TAPS * 4;
  if (smem < WAVES * 2 * bn * 4) smem = WAVES * 2 * bn * 4;
  const int ntile_n = d->cout_pad / bn;
  int per_cu = (160 * 1024) / (smem + 1024);
  // ONE 8-wave workgroup per CU: measured on the 1x1 layers of the step (64..384 channels at 64x128 / 128x256) 256 workgroups beat
  // 512 by 15-30 % -- half the weight-panel fills and half the same-address statistics atomics, two tiles per wave to pipeline
  if (per_cu > 1) per_cu = 1;
  if (g_stream_per_cu > 0) per_cu = g_stream_per_cu;
  if (per_cu < 1) per_cu = 1;
  int per_xcd = 32 * per_cu / ntile_n;                 // 32 CUs per XCD
  if (per_xcd < 1) per_xcd = 1;
  const int need = (k.tiles_per_xcd + WAVES - 1) / WAVES;
  if (per_xcd > need) per_xcd = need;
  const int grid_x = per_xcd * 8;
  hipStream_t st = (hipStream_t)stream;
  const bool kb2 = (d->cin_pad % 64) == 0;
  if (bn == 96) return kb2 ? launch<96, 2>(k, grid_x, ntile_n, smem, st) : launch<96, 1>(k, grid_x, ntile_n, smem, st);
  if (bn == 64) return kb2 ? launch<64, 2>(k, grid_x, ntile_n, smem, st) : launch<64, 1>(k, grid_x, ntile_n, smem, st);
  return kb2 ? launch<32, 2>(k, grid_x, ntile_n, smem, st) : launch<32, 1>(k, grid_x, ntile_n, smem, st);
}
