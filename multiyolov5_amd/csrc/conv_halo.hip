// LDS-staged input tiles for the k x k (3x3, dilated 3x3, 5x5) stride-1 convolutions and their dgrads (fp16).
//
//   The streaming kernel (conv_stream.hip) re-issues the global loads of the same pixel rows once per tap: fine for 1x1, but a 3x3
//   layer pulls every input byte nine times through the vector-memory path (64->64 @64x128 ran at 0.08 of its roofline).
//   Here a workgroup (4 waves) owns an output tile of TH x 32 pixels of one image and
//     * stages the tile's input HALO ((TH+kh-1) x (32+kw-1) pixels) through LDS once per 32-channel chunk, double buffered: the
//       global loads of chunk s+1 are in flight (registers) while chunk s is computed, one barrier per chunk;
//     * serves all taps from LDS: an MFMA B fragment (16 consecutive pixels of one halo row, 16-byte K segment lane>>4) is one
//       ds_read_b128 at (pixel + tap offset); halo pixels are 64-byte rows whose 16-byte segment index is XOR-ed with
//       2*bit2(pixel), which is conflict-free for the ds_read_b128 lane groups at EVERY pixel alignment (the +-1 / +-d tap shifts);
//     * keeps the whole weight panel of its N tile ([BN][taps*cin_pad]) in LDS for the life of the (persistent) workgroup;
//     * issues the MFMA as D^T = W . X^T over a row-permuted weight panel (panel_chan, myolo_dev.h: lane = pixel, 8 consecutive
//       channels per lane across a fragment pair): 16-byte NHWC stores from registers, BatchNorm statistics accumulated per lane and
//       flushed once per workgroup.
//   Out-of-image halo pixels and padded channels are masked by the buffer-resource bounds check (no branches, no zero page).
//   Tiles are dealt in XCD-contiguous ranges (neighbouring tiles share halo rows in one L2).
//
// Same contract as myolo_conv (include/myolo.h); selected by myolo_conv when the layer qualifies.
#include "myolo_dev.h"
#include <stdlib.h>
#include <string.h>

namespace halo {

constexpr int THREADS = 256;
constexpr int TW = 32;
constexpr int KCH = 32;             // halves per chunk = one 64-byte LDS pixel row

struct ConvH {
  const char* x; int64_t x_sn, x_sh, x_sw; int Hi, Wi, Cin;
  char* y; int64_t y_sn, y_sh, y_sw; int Ho, Wo, Cout, N;
  const char* w; int cin_pad, cout_pad, wtaps, ntaps;
  int tap_off[MYOLO_MAX_TAPS], tap_w[MYOLO_MAX_TAPS];      // halo-relative pixel offset (dy+oy0)*hw + (dx+ox0)
  const float* scale; const float* shift; int act; int accumulate;
  const char* res; int64_t r_sn, r_sh, r_sw;
  float* stats;
  int tiles_x, tiles_y, ntiles, tiles_per_xcd, pitchB;
  int hw, hh, ox0, oy0, nvec;        // halo width / height (pixels), origin offset (= -min dx, -min dy), 16-byte vectors per chunk
  int x_bytes, y_bytes, r_bytes;
  int xbuf_bytes;                    // bytes of one halo buffer
  int dbg;                           // profiling only (set_option "halo_dbg"): 1 no stores, 2 no global loads, 4 no MFMA, 8 no panel
  // S2 (fused dgrad of a stride-2 convolution): wave w owns output parity class (w>>1, w&1) with its own tap list
  int cls_ntaps[4], cls_off[4][4], cls_w[4][4];
  int Hfull, Wfull;                  // extent of the full gradient tensor (y_* describe it: strides of the whole tensor)
  BnbArgs bnb;                       // BatchNorm-backward statistics folded into the epilogue (BNS)
};

__device__ __forceinline__ int swzB(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }   // weight panel: H = {0,2,3,1}
__device__ __forceinline__ int xoff(int pix, int seg) { return pix * 64 + ((seg ^ ((pix >> 1) & 2)) << 4); }   // halo tile

// MF: 16-pixel fragments per wave (4: tile 8 x 32, 2: tile 4 x 32); NVT: staging vectors per thread and chunk
// EPI 0: raw output (+ BatchNorm statistics); EPI 1: scale/shift + activation (eval).  EXTRA: residual / accumulate loads.
// S2 = 1: the dgrad of a STRIDE-2 convolution, all four output parities in one launch.  gx[2a+py][2b+px] = sum over the taps of parity
//   class (py,px) of dy[a+oy_t][b+ox_t] . W^T[t] (1, 2, 2 and 4 taps for a 3x3): the output tile is still 8 x 32 pixels of gx, wave w
//   owns parity (w>>1, w&1) = 4 rows x 16 pixels, the dy halo is 5 x 17 pixels per 32-channel chunk, every gx row is written once
//   with all its pixels (the per-parity launches of round 1 each re-read dy and wrote every other 64-byte pixel of gx).
// BNS = 1 (with EPI 0): the stored gradient completes gout of a BatchNorm layer -> its backward sums (myolo_conv_desc.bnb)
template <int BN, int MF, int NVT, int EPI, int EXTRA, int S2 = 0, int BNS = 0>
__global__ __launch_bounds__(THREADS) void conv_halo_kernel(const ConvH p) {
  constexpr int NF = BN / 16;
  constexpr int TH = MF * 2;          // 4 waves x (MF/2) rows x 2 fragments per row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sB = smem;                                                        // [BN][pitchB] weight panel
  float* sT = reinterpret_cast<float*>(smem + (size_t)BN * p.pitchB);     // [4][BN]: scale, shift (EPI 1) | mean, invstd, sc, sh (BNS)
  int* sTap = reinterpret_cast<int*>(sT + 4 * BN);                        // [MAX_TAPS] halo pixel offset of each tap
  char* sX = reinterpret_cast<char*>(sTap + 32);                          // [2][xbuf_bytes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int tn = blockIdx.y;

  // ---- epilogue constants / tap table, once per workgroup (the weight panel is staged below, behind the first halo loads) ----
  if (EPI == 1)
    for (int c = tid; c < BN; c += THREADS) {
      const int cg = tn * BN + c;
      sT[c] = (p.scale && cg < p.Cout) ? p.scale[cg] : 1.0f;
      sT[BN + c] = (p.shift && cg < p.Cout) ? p.shift[cg] : 0.0f;
    }
  int sgi = -1;                          // BNS: the segment this N tile belongs to (segments are multiples of BN wide)
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
  if (S2) { if (tid < 16) { sTap[tid] = p.cls_off[tid >> 2][tid & 3]; sTap[16 + tid] = p.cls_w[tid >> 2][tid & 3]; } }
  else for (int t = tid; t < p.ntaps; t += THREADS) sTap[t] = p.tap_off[t];

  constexpr int OOB = 0x7fff0000;
  const bool half_tail = (p.Cout & 4) != 0;      // wave-uniform: the last 8-channel group of the layer is half a group
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.res ? p.res : p.y), 0, p.res ? p.r_bytes : 0, 0x00020000);

  // ---- staging roles: vector v = tid + j*256 of a chunk = (halo pixel v>>2, 16-byte segment v&3) ----
  int s_rel[NVT], s_pyx[NVT], s_lds[NVT];
#pragma unroll
  for (int j = 0; j < NVT; ++j) {
    const int v = tid + j * THREADS;
    const int pv = v >> 2, seg = v & 3;
    const int py = pv / p.hw, px = pv - py * p.hw;
    const bool in = v < p.nvec;
    s_pyx[j] = in ? ((py << 16) | px) : -1;
    s_rel[j] = (py * (int)p.x_sh + px * (int)p.x_sw) * 2 + seg * 16;
    s_lds[j] = xoff(pv, seg);
  }
  const int seg_c = (tid & 3) * 8;      // first channel of this thread's segment inside a chunk

  // ---- tiles of this workgroup: XCD-contiguous ranges ----
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, sstride = gridDim.x >> 3;
  const int tile_lo = xcd * p.tiles_per_xcd;
  int tile_hi = tile_lo + p.tiles_per_xcd;
  if (tile_hi > p.ntiles) tile_hi = p.ntiles;
  const int my_first = tile_lo + slot;
  const int ntl = my_first < tile_hi ? (tile_hi - my_first + sstride - 1) / sstride : 0;
  const int kchunks = p.cin_pad / KCH;
  const int nsteps = ntl * kchunks;
  const int tiles_per_img = p.tiles_x * p.tiles_y;

  // issue cursor
  int i_tile = my_first, i_kc = 0;
  int i_base = 0, i_mask = 0;           // byte offset of the halo origin of the tile; per-vector validity bits
  auto decode_issue = [&]() {
    const int n = i_tile / tiles_per_img; const int r = i_tile - n * tiles_per_img;
    const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
    const int hy0 = S2 ? ty * (TH / 2) - p.oy0 : ty * TH - p.oy0, hx0 = S2 ? tx * (TW / 2) - p.ox0 : tx * TW - p.ox0;
    i_base = (n * (int)p.x_sn + hy0 * (int)p.x_sh + hx0 * (int)p.x_sw) * 2;
    i_mask = 0;
#pragma unroll
    for (int j = 0; j < NVT; ++j) {
      const int py = s_pyx[j] >> 16, px = s_pyx[j] & 0xffff;
      const bool ok = s_pyx[j] >= 0 && (unsigned)(hy0 + py) < (unsigned)p.Hi && (unsigned)(hx0 + px) < (unsigned)p.Wi;
      i_mask |= ok ? (1 << j) : 0;
    }
  };
  uint4 stage[NVT];
  auto issue = [&]() {
    const int cb = i_kc * KCH;
    const bool cok = cb + seg_c < p.Cin;
#pragma unroll
    for (int j = 0; j < NVT; ++j) {
      const bool ok = cok && ((i_mask >> j) & 1) && !(p.dbg & 2);
      const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? i_base + s_rel[j] + cb * 2 : OOB, 0, 0);
      stage[j] = uint4{v.x, v.y, v.z, v.w};
    }
    if (++i_kc == kchunks) { i_kc = 0; i_tile += sstride; }
  };
  auto commit = [&](int buf) {          // registers -> LDS halo buffer
    char* dst = sX + buf * p.xbuf_bytes;
#pragma unroll
    for (int j = 0; j < NVT; ++j)
      if (s_pyx[j] >= 0) *reinterpret_cast<uint4*>(dst + s_lds[j]) = stage[j];
  };

  // compute cursor
  int c_tile = my_first, c_kc = 0;
  f4_t acc[MF][NF];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};
  float st_s[NF][4], st_q[NF][4];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int r = 0; r < 4; ++r) { st_s[nf][r] = 0.f; st_q[nf][r] = 0.f; }

  // this lane's pixel inside the halo tile for each of its fragments, at tap offset 0
  int pbase[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) pbase[mf] = S2 ? mf * p.hw + l15 : (wave * (MF / 2) + (mf >> 1)) * p.hw + (mf & 1) * 16 + l15;

  auto epilogue = [&](int tile) {
    const int n = tile / tiles_per_img; const int r0 = tile - n * tiles_per_img;
    const int ty = r0 / p.tiles_x, tx = r0 - ty * p.tiles_x;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int oy = S2 ? ty * TH + 2 * mf + (wave >> 1) : ty * TH + wave * (MF / 2) + (mf >> 1);
      const int ox = S2 ? tx * TW + 2 * l15 + (wave & 1) : tx * TW + (mf & 1) * 16 + l15;
      const bool mok = (S2 ? (oy < p.Hfull && ox < p.Wfull) : (oy < p.Ho && ox < p.Wo)) && !(p.dbg & 1);
      const int yoff = (n * (int)p.y_sn + oy * (int)p.y_sh + ox * (int)p.y_sw) * 2;
      const int roff = EXTRA ? (n * (int)p.r_sn + oy * (int)p.r_sh + ox * (int)p.r_sw) * 2 : 0;
#pragma unroll
      for (int q = 0; q < NF / 2; ++q) {
        // permuted weight-panel rows (panel_chan, myolo_dev.h): fragments 2q, 2q+1 hold channels cl .. cl+7 of this lane's pixel
        const int cl = q * 32 + 8 * lq;
        const int c0 = tn * BN + cl;
        float v[8];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v0 = acc[mf][2 * q + h][r];
            acc[mf][2 * q + h][r] = 0.f;
            if (EPI == 0) { v[4 * h + r] = v0; if (mok && !BNS) { st_s[2 * q + h][r] += v0; st_q[2 * q + h][r] += v0 * v0; } }
            else v[4 * h + r] = act_f(v0 * sT[cl + 4 * h + r] + sT[BN + cl + 4 * h + r], p.act);
          }
        const bool ok8 = mok && c0 + 8 <= p.Cout, ok4 = mok && !ok8 && c0 + 4 <= p.Cout;     // Cout % 4 == 0 (host)
        if (EXTRA) {
          if (p.res) add_h8(v, buf_load_h8(rr, roff + c0 * 2, ok8, ok4, half_tail));
          if (p.accumulate) add_h8(v, buf_load_h8(ry, yoff + c0 * 2, ok8, ok4, half_tail));
        }
        const u32x4_t o = pack_h8(v);
        __builtin_amdgcn_raw_buffer_store_b128(o, ry, ok8 ? yoff + c0 * 2 : OOB, 0, 0);
        if (half_tail) __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{o.x, o.y}, ry, ok4 ? yoff + c0 * 2 : OOB, 0, 0);
        if (BNS) {
          if (sgi >= 0) {                // (wave-uniform) dz = gout * act'(z) of the normalised layer, from its raw output at this pixel
            const BnbSeg& sg = p.bnb.seg[sgi];
            const bool okb = ok8 && c0 < sg.c1;                    // segments start and end on N-tile boundaries (bnb_aligned)
            const char* yp = okb ? sg.y + ((int64_t)n * sg.y_sn + (int64_t)oy * sg.y_sh + (int64_t)ox * sg.y_sw + (c0 - sg.c0)) * 2 : zero_page();
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

  auto load_frags = [&](int t, const char* xb, uint4* fa, uint4* fb) {
    const int toff = S2 ? sTap[wave * 4 + t] : sTap[t];
    const int tw = S2 ? sTap[16 + wave * 4 + t] : t;           // S2: the panel holds all weight taps in natural order
    const char* brow = sB + l15 * p.pitchB + (tw * p.cin_pad + c_kc * KCH) * 2 + ((lq ^ swzB(l15)) << 4);
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) fb[nf] = *reinterpret_cast<const uint4*>(brow + nf * 16 * p.pitchB);
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) fa[mf] = *reinterpret_cast<const uint4*>(xb + xoff(pbase[mf] + toff, lq));
  };
  auto mma = [&](const uint4* fa, const uint4* fb) {
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)   // weights as the A operand, pixels as the B operand: D[cout][pixel]
        acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const h8_t*>(&fb[nf]),
                                                             *reinterpret_cast<const h8_t*>(&fa[mf]), acc[mf][nf], 0, 0, 0);
  };

  // ---- pipeline over (tile, chunk) steps ----
  if (nsteps > 0) {
    decode_issue();
    issue();                             // the first halo chunk is in flight while the weight panel is staged
  }
  if (!(p.dbg & 8)) stage_weight_panel<BN, THREADS, true>(sB, p.w, tn, p.pitchB, p.cin_pad, p.ntaps, p.wtaps, p.tap_w, tid);
  if (nsteps > 0) commit(0);
  __syncthreads();                       // weight panel, tables and the first halo chunk are in LDS
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    const bool more = s + 1 < nsteps;
    if (more) {
      if (i_kc == 0) decode_issue();     // (i_tile already advanced by the previous issue)
      issue();                           // global loads of step s+1: in flight during the MFMAs below
    }
    const char* xb = sX + buf * p.xbuf_bytes;
    uint4 fa0[MF], fb0[NF], fa1[MF], fb1[NF];
    const int nt = S2 ? __builtin_amdgcn_readfirstlane(p.cls_ntaps[wave]) : p.ntaps;
    load_frags(0, xb, fa0, fb0);
    int t = (p.dbg & 4) ? nt : 0;
    for (; t + 1 < nt; t += 2) {
      load_frags(t + 1, xb, fa1, fb1);
      mma(fa0, fb0);
      if (t + 2 < nt) load_frags(t + 2, xb, fa0, fb0);
      mma(fa1, fb1);
    }
    if ((nt & 1) && !(p.dbg & 4)) mma(fa0, fb0);
    if (++c_kc == kchunks) { c_kc = 0; epilogue(c_tile); c_tile += sstride; }
    if (more) commit(buf ^ 1);
    __syncthreads();
  }

  if (EPI == 0 && (BNS ? sgi >= 0 : p.stats != nullptr)) {
    float* red = reinterpret_cast<float*>(sX);         // [4 waves][2*BN]; the loop ended with a barrier
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
      const float a = red[t] + red[2 * BN + t] + red[4 * BN + t] + red[6 * BN + t];
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

template <int BN, int MF, int NVT, int EPI, int EXTRA, int S2 = 0, int BNS = 0>
int launch4(const ConvH& k, int grid_x, int ntile_n, int smem, hipStream_t st) {
  auto kern = conv_halo_kernel<BN, MF, NVT, EPI, EXTRA, S2, BNS>;
  MYOLO_ENSURE_DYN_SMEM(kern, smem);
  hipLaunchKernelGGL(kern, dim3(grid_x, ntile_n), dim3(THREADS), smem, st, k);
  MYOLO_CHECK_LAUNCH();
  return 0;
}
template <int BN, int MF, int NVT>
int launch3(const ConvH& k, int grid_x, int ntile_n, int smem, hipStream_t st) {
  const bool raw = !k.scale && !k.shift && k.act == MYOLO_ACT_NONE;
  const bool extra = k.res != nullptr || k.accumulate;
  if (raw && k.bnb.n > 0)
    return extra ? launch4<BN, MF, NVT, 0, 1, 0, 1>(k, grid_x, ntile_n, smem, st) : launch4<BN, MF, NVT, 0, 0, 0, 1>(k, grid_x, ntile_n, smem, st);
  if (raw) return extra ? launch4<BN, MF, NVT, 0, 1>(k, grid_x, ntile_n, smem, st) : launch4<BN, MF, NVT, 0, 0>(k, grid_x, ntile_n, smem, st);
  return extra ? launch4<BN, MF, NVT, 1, 1>(k, grid_x, ntile_n, smem, st) : launch4<BN, MF, NVT, 1, 0>(k, grid_x, ntile_n, smem, st);
}
template <int BN, int MF>
int launch2(const ConvH& k, int nvt, int grid_x, int ntile_n, int smem, hipStream_t st) {
  return nvt <= 6 ? launch3<BN, MF, 6>(k, grid_x, ntile_n, smem, st) : launch3<BN, MF, 10>(k, grid_x, ntile_n, smem, st);
}

}  // namespace halo

static inline int halo_panel_pitch(int K) {                    // bytes; multiple of 64 with an odd number of 64-byte blocks
  int blocks = (K * 2 + 63) / 64;
  if (!(blocks & 1)) ++blocks;
  return blocks * 64;
}

static int g_halo_off = -1;          // -1: from the environment (MYOLO_NO_HALO)
static int g_halo_dbg = 0;
static int g_halo_min_tiles = -1;    // minimum number of workgroup tiles (MYOLO_HALO_MIN_TILES, default 512 = two per CU: below that the
                                     // 64-row tiles of the LDS-tiled kernel fill the chip better -- 3x3 64->64 at 1x128x256: 23.0 vs 15.7 us, r3 A/B)

int myolo_conv_halo_set(const char* name, int value) {
  if (!strcmp(name, "halo_off")) { g_halo_off = value; return 0; }
  if (!strcmp(name, "halo_min_tiles")) { g_halo_min_tiles = value; return 0; }
  if (!strcmp(name, "halo_dbg")) { g_halo_dbg = value; return 0; }
  if (!strncmp(name, "small_", 6)) return myolo_conv_small_set(name, value);
  return myolo_wgrad_tile_set(name, value);        // "wgrad_tile_off"
}

// returns -1 when the layer does not qualify (caller falls back to the other kernels), else a hipError_t / 0
int myolo_conv_halo_try(const myolo_conv_desc* d, void* stream, int* bnb_done) {
  using namespace halo;
  *bnb_done = 0;
  if (g_halo_off < 0) g_halo_off = 0;                     // (myolo_set_option("halo_off" / "halo_min_tiles"): tests and sweeps)
  if (g_halo_min_tiles < 0) g_halo_min_tiles = 512;
  if (g_halo_off || d->x.dtype != MYOLO_F16 || d->det_no > 0 || (d->y.c & 3)) return -1;
  if (d->ntaps < 2 || d->stride != 1 || d->up_shift != 0 || d->cin_pad % KCH) return -1;
  if (d->x.h != d->y.h || d->x.w != d->y.w) return -1;
  if (d->stats && (d->scale || d->shift || d->act != MYOLO_ACT_NONE)) return -1;
  int mindy = 0, maxdy = 0, mindx = 0, maxdx = 0;
  for (int t = 0; t < d->ntaps; ++t) {
    mindy = d->tap_dy[t] < mindy ? d->tap_dy[t] : mindy; maxdy = d->tap_dy[t] > maxdy ? d->tap_dy[t] : maxdy;
    mindx = d->tap_dx[t] < mindx ? d->tap_dx[t] : mindx; maxdx = d->tap_dx[t] > maxdx ? d->tap_dx[t] : maxdx;
  }
  const int K = d->ntaps * d->cin_pad;
  // K > 832 (128-channel 3x3 and up): only a 32-wide N tile of the panel fits beside the halo buffers, the input would be staged
  // Cout/32 times and the LDS-tiled kernel measured faster (27 vs 31 us at 128->128, 32x64)
  constexpr int max_k = 832;
  if (K > max_k) return -1;
  const int pitch = halo_panel_pitch(K);
  const int64_t px_total = (int64_t)d->y.n * d->y.h * d->y.w;
  // tile height: 8 rows unless that leaves the chip short of workgroups
  int bn = 0, mf = 0, hh = 0, hw = 0, nvt = 0, xbuf = 0, smem = 0;
  const int bns[2] = {64, 32};
  for (int bi = 0; bi < 2 && !bn; ++bi) {
    const int b = bns[bi];
    if (d->cout_pad % b) continue;
    for (int m = 4; m >= 2 && !bn; m -= 2) {
      const int th = m * 2;
      const int64_t tiles = (int64_t)d->y.n * ((d->y.h + th - 1) / th) * ((d->y.w + TW - 1) / TW) * (d->cout_pad / b);
      if (m == 4 && tiles < 384 && d->y.h > 4) continue;            // < 1.5 workgroups per CU: use 4-row tiles
      const int hh_ = th + (maxdy - mindy), hw_ = TW + (maxdx - mindx);
      const int nvec = hh_ * hw_ * 4;
      const int nvt_ = (nvec + THREADS - 1) / THREADS;
      if (nvt_ > 10) continue;
      const int xb = hh_ * hw_ * 64;
      const int sm = b * pitch + 4 * b * 4 + 32 * 4 + 2 * xb;
      if (sm > 160 * 1024) continue;
      bn = b; mf = m; hh = hh_; hw = hw_; nvt = nvt_; xbuf = xb; smem = sm;
    }
  }
  if (!bn) return -1;
  ConvH k;
  k.x = (const char*)d->x.ptr; k.x_sn = d->x.sn; k.x_sh = d->x.sh; k.x_sw = d->x.sw;
  k.Hi = d->x.h; k.Wi = d->x.w; k.Cin = d->x.c;
  k.y = (char*)d->y.ptr; k.y_sn = d->y.sn; k.y_sh = d->y.sh; k.y_sw = d->y.sw;
  k.Ho = d->y.h; k.Wo = d->y.w; k.Cout = d->y.c; k.N = d->y.n;
  k.w = (const char*)d->w; k.cin_pad = d->cin_pad; k.cout_pad = d->cout_pad; k.wtaps = d->wtaps; k.ntaps = d->ntaps;
  k.ox0 = -mindx; k.oy0 = -mindy; k.hw = hw; k.hh = hh; k.nvec = hh * hw * 4; k.xbuf_bytes = xbuf;
  for (int i = 0; i < MYOLO_MAX_TAPS; ++i) {
    k.tap_off[i] = i < d->ntaps ? (d->tap_dy[i] - mindy) * hw + (d->tap_dx[i] - mindx) : 0;
    k.tap_w[i] = d->tap_w[i];
  }
  k.scale = d->scale; k.shift = d->shift; k.act = d->act; k.accumulate = d->accumulate;
  k.res = (const char*)d->res.ptr; k.r_sn = d->res.sn; k.r_sh = d->res.sh; k.r_sw = d->res.sw;
  k.stats = d->stats; k.dbg = g_halo_dbg;
  k.bnb.n = 0;
  if (d->bnb && d->nbnb > 0 && !d->stats && !d->scale && !d->shift && d->act == MYOLO_ACT_NONE && bnb_aligned(d, bn)) { bnb_fill(&k.bnb, d); *bnb_done = 1; }
  const int th = mf * 2;
  k.tiles_x = (k.Wo + TW - 1) / TW; k.tiles_y = (k.Ho + th - 1) / th;
  const int64_t nt = (int64_t)k.N * k.tiles_x * k.tiles_y;
  if (px_total <= 0 || nt > 0x3fffffff) return MYOLO_EINVAL;
  k.ntiles = (int)nt;
  if (k.ntiles * (d->cout_pad / bn) < g_halo_min_tiles) return -1;     // tiny maps: launch-latency bound either way
  k.tiles_per_xcd = (k.ntiles + 7) / 8;
  k.pitchB = pitch;
  auto span = [](const myolo_tensor& t) -> int64_t {
    return (((int64_t)t.n - 1) * t.sn + ((int64_t)t.h - 1) * t.sh + ((int64_t)t.w - 1) * t.sw + t.c) * 2;
  };
  // halo origins lie up to (oy0 rows + ox0 pixels) before the view: offsets are formed in 32-bit signed arithmetic
  const int64_t xb = span(d->x), yb = span(d->y), rb = d->res.ptr ? span(d->res) : 0;
  if (xb >= 0x3ffe0000LL || yb >= 0x3ffe0000LL || rb >= 0x3ffe0000LL) return -1;
  k.x_bytes = (int)xb; k.y_bytes = (int)yb; k.r_bytes = (int)rb;
  const int ntile_n = d->cout_pad / bn;
  int per_cu = (160 * 1024) / (smem + 512);
  if (per_cu > 2) per_cu = 2;
  if (per_cu < 1) per_cu = 1;
  int per_xcd = 32 * per_cu / ntile_n;
  if (per_xcd < 1) per_xcd = 1;
  if (per_xcd > k.tiles_per_xcd) per_xcd = k.tiles_per_xcd;
  const int grid_x = per_xcd * 8;
  hipStream_t st = (hipStream_t)stream;
  if (bn == 64) return mf == 4 ? launch2<64, 4>(k, nvt, grid_x, ntile_n, smem, st) : launch2<64, 2>(k, nvt, grid_x, ntile_n, smem, st);
  return mf == 4 ? launch2<32, 4>(k, nvt, grid_x, ntile_n, smem, st) : launch2<32, 2>(k, nvt, grid_x, ntile_n, smem, st);
}

// ---- fused dgrad of a stride-2 convolution: the four parity sub-convolutions the host derives (engine.taps_dgrad) in ONE launch ----
static int halo_s2_try(const myolo_conv_desc* const* d4, void* stream, int* bnb_done, myolo_tensor* full) {
  using namespace halo;
  *bnb_done = 0;
  if (g_halo_off < 0) g_halo_off = 0;
  constexpr int s2_off = 0;
  if (g_halo_off || s2_off) return -1;
  const myolo_conv_desc* d0 = d4[0];
  if (!d0 || d0->x.dtype != MYOLO_F16 || d0->y.dtype != MYOLO_F16 || (d0->y.c & 3) || d0->cin_pad % KCH) return -1;
  if ((d0->y.sh & 1) || (d0->y.sw & 1)) return -1;
  const int64_t fsh = d0->y.sh / 2, fsw = d0->y.sw / 2;
  int minoy = 1 << 20, maxoy = -(1 << 20), minox = 1 << 20, maxox = -(1 << 20);
  for (int k = 0; k < 4; ++k) {
    const myolo_conv_desc* d = d4[k];
    if (!d || !d->x.ptr || !d->y.ptr || !d->w) return -1;
    if (d->x.ptr != d0->x.ptr || d->x.n != d0->x.n || d->x.h != d0->x.h || d->x.w != d0->x.w || d->x.c != d0->x.c ||
        d->x.sn != d0->x.sn || d->x.sh != d0->x.sh || d->x.sw != d0->x.sw || d->x.dtype != MYOLO_F16)
      return -1;
    if (d->w != d0->w || d->cin_pad != d0->cin_pad || d->cout_pad != d0->cout_pad || d->wtaps != d0->wtaps) return -1;
    if (d->stride != 1 || d->up_shift != 0 || d->ntaps < 1 || d->ntaps > 4 || d->det_no > 0) return -1;
    if (d->scale || d->shift || d->act != MYOLO_ACT_NONE || d->res.ptr || d->stats || d->accumulate != d0->accumulate) return -1;
    const int py = k >> 1, px = k & 1;
    if (d->y.dtype != MYOLO_F16 || d->y.c != d0->y.c || d->y.n != d0->y.n || d->y.sn != d0->y.sn || d->y.sh != d0->y.sh || d->y.sw != d0->y.sw)
      return -1;
    if ((const char*)d->y.ptr != (const char*)d0->y.ptr + (py * fsh + px * fsw) * 2) return -1;
    for (int t = 0; t < d->ntaps; ++t) {
      minoy = d->tap_dy[t] < minoy ? d->tap_dy[t] : minoy; maxoy = d->tap_dy[t] > maxoy ? d->tap_dy[t] : maxoy;
      minox = d->tap_dx[t] < minox ? d->tap_dx[t] : minox; maxox = d->tap_dx[t] > maxox ? d->tap_dx[t] : maxox;
      if (d->tap_w[t] < 0 || d->tap_w[t] >= d->wtaps) return -1;
    }
  }
  const int Hfull = d4[0]->y.h + d4[2]->y.h, Wfull = d4[0]->y.w + d4[1]->y.w;
  if (d4[1]->y.h != d4[0]->y.h || d4[3]->y.h != d4[2]->y.h || d4[2]->y.w != d4[0]->y.w || d4[3]->y.w != d4[1]->y.w) return -1;
  if (d4[0]->y.h < d4[2]->y.h || d4[0]->y.w < d4[1]->y.w) return -1;
  const int hh = 4 + (maxoy - minoy), hw = 16 + (maxox - minox);
  const int nvec = hh * hw * 4;
  if (nvec > 6 * THREADS) return -1;
  const int K = d0->wtaps * d0->cin_pad;
  // K = 2304 (256 gradient channels): only a 32-wide N tile of the panel fits, dy is staged Cout/32 times and four LDS-tiled launches
  // measured faster (133 vs 98 us at 16x32x64x256 -> 128, 81 vs 76 us at 16x16x32x256 -> 256; K = 1152: 86 vs 104 us the other way)
  constexpr int s2_max_k = 1152;
  if (K > s2_max_k) return -1;
  const int pitch = halo_panel_pitch(K);
  const int xbuf = hh * hw * 64;
  int bn = 0, smem = 0;
  const int bns[2] = {64, 32};
  for (int bi = 0; bi < 2 && !bn; ++bi) {
    if (d0->cout_pad % bns[bi]) continue;
    const int sm = bns[bi] * pitch + 4 * bns[bi] * 4 + 32 * 4 + 2 * xbuf;
    if (sm <= 160 * 1024) { bn = bns[bi]; smem = sm; }
  }
  if (!bn) return -1;
  ConvH k;
  memset(&k, 0, sizeof(k));
  k.x = (const char*)d0->x.ptr; k.x_sn = d0->x.sn; k.x_sh = d0->x.sh; k.x_sw = d0->x.sw;
  k.Hi = d0->x.h; k.Wi = d0->x.w; k.Cin = d0->x.c;
  k.y = (char*)d0->y.ptr; k.y_sn = d0->y.sn; k.y_sh = fsh; k.y_sw = fsw;
  k.Ho = Hfull; k.Wo = Wfull; k.Hfull = Hfull; k.Wfull = Wfull; k.Cout = d0->y.c; k.N = d0->y.n;
  k.w = (const char*)d0->w; k.cin_pad = d0->cin_pad; k.cout_pad = d0->cout_pad; k.wtaps = d0->wtaps; k.ntaps = d0->wtaps;
  for (int t = 0; t < MYOLO_MAX_TAPS; ++t) { k.tap_off[t] = 0; k.tap_w[t] = t < d0->wtaps ? t : 0; }      // panel: every weight tap, natural order
  for (int c = 0; c < 4; ++c) {
    k.cls_ntaps[c] = d4[c]->ntaps;
    for (int t = 0; t < 4; ++t) {
      const bool in = t < d4[c]->ntaps;
      k.cls_off[c][t] = in ? (d4[c]->tap_dy[t] - minoy) * hw + (d4[c]->tap_dx[t] - minox) : 0;
      k.cls_w[c][t] = in ? d4[c]->tap_w[t] : 0;
    }
  }
  k.ox0 = -minox; k.oy0 = -minoy; k.hw = hw; k.hh = hh; k.nvec = nvec; k.xbuf_bytes = xbuf;
  k.act = MYOLO_ACT_NONE; k.accumulate = d0->accumulate; k.dbg = g_halo_dbg;
  k.tiles_x = (Wfull + TW - 1) / TW; k.tiles_y = (Hfull + 7) / 8;
  const int64_t nt = (int64_t)k.N * k.tiles_x * k.tiles_y;
  if (nt <= 0 || nt > 0x3fffffff) return -1;
  k.ntiles = (int)nt;
  k.tiles_per_xcd = (k.ntiles + 7) / 8;
  k.pitchB = pitch;
  const int64_t xb = (((int64_t)d0->x.n - 1) * d0->x.sn + ((int64_t)d0->x.h - 1) * d0->x.sh + ((int64_t)d0->x.w - 1) * d0->x.sw + d0->x.c) * 2;
  const int64_t yb = (((int64_t)k.N - 1) * k.y_sn + ((int64_t)Hfull - 1) * fsh + ((int64_t)Wfull - 1) * fsw + k.Cout) * 2;
  if (xb >= 0x3ffe0000LL || yb >= 0x3ffe0000LL) return -1;
  k.x_bytes = (int)xb; k.y_bytes = (int)yb; k.r_bytes = 0;
  const int ntile_n = d0->cout_pad / bn;
  int per_cu = (160 * 1024) / (smem + 512);
  if (per_cu > 2) per_cu = 2;
  if (per_cu < 1) per_cu = 1;
  int per_xcd = 32 * per_cu / ntile_n;
  if (per_xcd < 1) per_xcd = 1;
  if (per_xcd > k.tiles_per_xcd) per_xcd = k.tiles_per_xcd;
  const int grid_x = per_xcd * 8;
  hipStream_t st = (hipStream_t)stream;
  if (d0->bnb && d0->nbnb > 0) {
    myolo_conv_desc tmp = *d0;               // (bnb channel ranges refer to the full gradient tensor: same channels in every parity)
    tmp.y.c = d0->y.c;
    if (bnb_aligned(&tmp, bn)) { bnb_fill(&k.bnb, d0); *bnb_done = 1; }
  }
  if (k.bnb.n > 0) {
    if (bn == 64)
      return k.accumulate ? launch4<64, 4, 6, 0, 1, 1, 1>(k, grid_x, ntile_n, smem, st) : launch4<64, 4, 6, 0, 0, 1, 1>(k, grid_x, ntile_n, smem, st);
    return k.accumulate ? launch4<32, 4, 6, 0, 1, 1, 1>(k, grid_x, ntile_n, smem, st) : launch4<32, 4, 6, 0, 0, 1, 1>(k, grid_x, ntile_n, smem, st);
  }
  if (bn == 64)
    return k.accumulate ? launch4<64, 4, 6, 0, 1, 1>(k, grid_x, ntile_n, smem, st) : launch4<64, 4, 6, 0, 0, 1>(k, grid_x, ntile_n, smem, st);
  return k.accumulate ? launch4<32, 4, 6, 0, 1, 1>(k, grid_x, ntile_n, smem, st) : launch4<32, 4, 6, 0, 0, 1>(k, grid_x, ntile_n, smem, st);
}

extern "C" int myolo_conv_dgrad_s2(const myolo_conv_desc* const* parity, int n, void* stream) {
  if (!parity || n < 1 || n > 4) return MYOLO_EINVAL;
  const myolo_conv_desc* d0 = parity[0];
  const bool want_bnb = d0 && d0->bnb && d0->nbnb > 0;
  // the whole gradient tensor the parity views interleave (for the statistics fallback)
  myolo_tensor full = d0 ? d0->y : myolo_tensor{};
  if (d0 && n == 4 && parity[1] && parity[2]) {
    full.h = parity[0]->y.h + parity[2]->y.h; full.w = parity[0]->y.w + parity[1]->y.w;
    full.sh = d0->y.sh / 2; full.sw = d0->y.sw / 2;
  }
  if (n == 4) {
    int done = 0;
    const int r = halo_s2_try(parity, stream, &done, &full);
    if (r != -1) {
      if (r) return r;
      return (want_bnb && !done) ? myolo_bnb_fallback(d0, &full, stream) : 0;
    }
  }
  for (int i = 0; i < n; ++i) {               // not fusable (fp32, big K, odd layout ...): the sub-convolutions one by one
    if (!parity[i]) return MYOLO_EINVAL;
    myolo_conv_desc sub = *parity[i];
    sub.bnb = nullptr; sub.nbnb = 0;          // (a parity launch sees a quarter of the pixels: the statistics follow below)
    const int r = myolo_conv(&sub, stream);
    if (r) return r;
  }
  if (want_bnb) {
    if (n != 4) return MYOLO_EINVAL;          // statistics need the complete tensor
    return myolo_bnb_fallback(d0, &full, stream);
  }
  return 0;
}
