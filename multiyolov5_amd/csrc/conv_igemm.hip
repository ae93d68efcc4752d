// Implicit-GEMM convolution for gfx950 (CDNA4): NHWC activations, packed [cout][tap][cin] weights,
// MFMA 16x16x32 f16 (or 16x16x4 f32 in parity mode), fp32 accumulate, fused epilogue.
//
//   GEMM view: M = N*Ho*Wo pixels, N = Cout, K = ntaps*Cin.   One workgroup = 256 threads = 4 waves
//   computes a BM(128) x BN(32|64|128) output tile; wave w owns rows [32w,32w+32) x all BN columns
//   (2 x BN/16 accumulator fragments).  The K loop walks (tap, 64-byte channel chunk); each step the
//   A tile (128 pixel rows x 64 B, gathered at the tap's shifted pixel) and the B tile (BN weight rows
//   x 64 B) are staged global -> registers -> LDS (double buffered, one barrier per step; the global
//   loads of step s+1 are in flight while step s computes).  LDS rows are 64 B with the 16-byte
//   segment index XOR-ed by H[(row>>2)&3] = {0,2,3,1}: conflict-free for the ds_read_b128 lane groups
//   of the MFMA fragment pattern (row = lane&15, segment = lane>>4) and for the staging writes.
//
//   Workgroups are persistent over M tiles (grid.x, XCD-contiguous ranges so that vertically adjacent
//   tiles of a 3x3 conv share one XCD's L2) with a fixed N tile (grid.y): the training-mode BatchNorm
//   statistics (sum, sum of squares of the raw fp32 accumulators) are kept in registers across the
//   tiles and flushed once per workgroup with 2*BN atomics.
//
//   Epilogue: v = acc*scale[c]+shift[c] -> act -> staged through LDS so that global stores are
//   16-byte row-contiguous (NHWC) -> (+ residual / += y) -> store, or Detect's permuted layout.
//
// Replaces: nn.Conv2d+BatchNorm2d+SiLU (reference models/common.py:34-46, 481-490), Detect.m (yolo.py:211-214),
// and their dgrad (transposed weights, flipped taps, stride-2 by output parity).
#include "myolo_dev.h"
#include <string.h>
#include <stdlib.h>

namespace {

constexpr int THREADS = 256;

struct ConvK {
  const char* x; int64_t x_sn, x_sh, x_sw; int Hi, Wi, Cin;
  char* y; int64_t y_sn, y_sh, y_sw; int Ho, Wo, Cout, N;
  const char* w; int cin_pad, cout_pad, wtaps, ntaps, stride, up;
  int tap_dy[MYOLO_MAX_TAPS], tap_dx[MYOLO_MAX_TAPS], tap_w[MYOLO_MAX_TAPS];
  const float* scale; const float* shift; int act; int accumulate;
  const char* res; int64_t r_sn, r_sh, r_sw;
  float* stats; int det_no; int M; int ntile_m; int tiles_per_xcd; int bm;
  BnbArgs bnb; int bn_off;           // BNS: statistics segments; byte offset of the [4][BN] constant table / reduction area in LDS
};

}  // namespace

// H = {0,2,3,1} packed two bits per entry: 0b01'11'10'00 = 0x78
namespace {
__device__ __forceinline__ int swz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }
__device__ __forceinline__ int lds_off2(int row, int seg) { return row * 64 + ((seg ^ swz(row)) << 4); }

template <typename T> struct Mma;
template <> struct Mma<half_t> {
  __device__ static void run(const uint4& a, const uint4& b, f4_t& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const h8_t*>(&a),
                                                 *reinterpret_cast<const h8_t*>(&b), acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  __device__ static void run(const uint4& a, const uint4& b, f4_t& acc) {
    const float* fa = reinterpret_cast<const float*>(&a);
    const float* fb = reinterpret_cast<const float*>(&b);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j], fb[j], acc, 0, 0, 0);
  }
};

// BM = 128 (two 16-row fragments per wave) or 64 (one): small maps get twice the workgroups, so that two latency-bound K loops
// share each CU instead of one
// KS = 64-byte K sub-chunks per step (1 or 2): a 128-byte step halves the per-step overhead (barrier, ~110 scalar/vector
// address instructions) per MFMA; each sub-chunk is its own swizzled [rows][64 B] plane in LDS.
// BNS = 1: the stored gradient completes gout of a BatchNorm layer -> its backward sums (myolo_conv_desc.bnb), computed on the final
// (accumulated, storage-rounded) values in the store loop; thread t always serves channel vector t % (BN/SEG)
template <typename T, int BN, int BM, int KS, int BNS = 0>
__global__ __launch_bounds__(THREADS) void conv_igemm_kernel(const ConvK p) {
  constexpr int MF = BM / 64;        // 16-row fragments per wave; also A rows staged per thread
  constexpr int SEG = ET<T>::SEG;   // elements per 16 B
  constexpr int KC = ET<T>::KC;     // elements per 64 B K-chunk
  constexpr int NF = BN / 16;
  constexpr int ES = (int)sizeof(T);
  constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64;
  constexpr int A_BYTES = A_PLANE * KS, B_BYTES = B_PLANE * KS;
  constexpr int CROW = BN * ES + 16;            // C staging row pitch (bytes)
  constexpr int BROWS = (BN + 63) / 64;         // B rows per thread (64 rows per pass)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                 // [2][A_BYTES]
  char* sB = smem + 2 * A_BYTES;   // [2][B_BYTES]
  char* sC = smem;                 // epilogue staging (aliases A/B)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tn = blockIdx.y;
  const int xcd = blockIdx.x & 7, bslot = blockIdx.x >> 3, bstride = gridDim.x >> 3;
  const int kchunks = p.cin_pad / (KC * KS);
  const int nsteps = p.ntaps * kchunks;
  const int HWo = p.Ho * p.Wo;
  const int Hlog = p.Hi << p.up, Wlog = p.Wi << p.up;

  // staging roles
  const int lrow = tid >> 2, lseg = tid & 3;   // A rows lrow, lrow+64 ; B rows lrow (+64)

  // per-lane BN statistics partials (persist across this workgroup's tiles)
  float st_s[NF], st_q[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) { st_s[i] = 0.f; st_q[i] = 0.f; }

  // per-lane epilogue constants: channel = tn*BN + nf*16 + (lane&15)
  float e_scale[NF], e_shift[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int c = tn * BN + nf * 16 + (lane & 15);
    e_scale[nf] = (p.scale && c < p.Cout) ? p.scale[c] : 1.0f;
    e_shift[nf] = (p.shift && c < p.Cout) ? p.shift[c] : 0.0f;
  }

  const char* wbase = p.w + (int64_t)(tn * BN) * p.wtaps * p.cin_pad * ES;

  int sgi = -1;
  float* sBN = reinterpret_cast<float*>(smem + p.bn_off);       // [4][BN] mean, invstd, sc, sh (outside the A/B/C staging area)
  float bs0[SEG], bs1[SEG];
#pragma unroll
  for (int i = 0; i < SEG; ++i) { bs0[i] = 0.f; bs1[i] = 0.f; }
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
        sBN[c] = mean; sBN[BN + c] = istd; sBN[2 * BN + c] = sc; sBN[3 * BN + c] = in ? sg.beta[ci] - mean * sc : 0.f;
      }
    }
    // (visible to the store loop: every tile's epilogue starts with a barrier)
  }

  for (int tslot = bslot; tslot < p.tiles_per_xcd; tslot += bstride) {
    const int tm = xcd * p.tiles_per_xcd + tslot;
    if (tm >= p.ntile_m) break;
    const int m0 = tm * BM;

    // decode this thread's two A rows
    int rn[MF], roy[MF], rox[MF]; bool rvalid[MF];
#pragma unroll
    for (int r = 0; r < MF; ++r) {
      const int m = m0 + lrow + r * 64;
      rvalid[r] = m < p.M;
      const int mm = rvalid[r] ? m : 0;
      rn[r] = mm / HWo;
      const int rem = mm - rn[r] * HWo;
      roy[r] = rem / p.Wo;
      rox[r] = rem - roy[r] * p.Wo;
    }

    f4_t acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

    // Register ring: the global loads of K step s+RING-1 are issued while step s computes, so RING-1 steps (not one) of
    // load latency are in flight per wave.  Small maps run ~1 workgroup per CU: with a single step of prefetch every K step
    // exposed a full L2 round trip (~0.9 us per 64-byte step, MFMA busy ~10 %).
    constexpr int RING = KS == 2 ? 3 : 4;
    uint4 ra[RING][MF * KS], rb[RING][BROWS * KS];
    const char* arow[MF];
#pragma unroll
    for (int r = 0; r < MF; ++r) arow[r] = nullptr;
    int cur_tap = -1, i_tap = 0, i_kc = 0;           // issue cursor (sequential over steps)

    auto issue_loads = [&](uint4* da, uint4* db, const bool live) {
      if (i_tap != cur_tap) {
        cur_tap = i_tap;
        const int tt = i_tap < p.ntaps ? i_tap : p.ntaps - 1;
        const int dy = p.tap_dy[tt], dx = p.tap_dx[tt];
#pragma unroll
        for (int r = 0; r < MF; ++r) {
          int iy = roy[r] * p.stride + dy, ix = rox[r] * p.stride + dx;
          const bool ok = rvalid[r] && iy >= 0 && iy < Hlog && ix >= 0 && ix < Wlog;
          iy >>= p.up; ix >>= p.up;
          arow[r] = ok ? p.x + ((int64_t)rn[r] * p.x_sn + (int64_t)iy * p.x_sh + (int64_t)ix * p.x_sw) * ES : nullptr;
        }
      }
      const int wt = p.tap_w[i_tap < p.ntaps ? i_tap : p.ntaps - 1];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int c0 = (i_kc * KS + ks) * KC + lseg * SEG;
        // unconditional loads, out-of-image / padded-channel / past-the-end lanes read the zero page (myolo_dev.h)
#pragma unroll
        for (int r = 0; r < MF; ++r) {
          const char* ap = (live && arow[r] != nullptr && c0 < p.Cin) ? arow[r] + (int64_t)c0 * ES : zero_page();
          da[ks * MF + r] = ldg16(ap);
        }
#pragma unroll
        for (int r = 0; r < BROWS; ++r) {
          int brow = lrow + r * 64;
          if (BN % 64) brow = brow < BN ? brow : brow - 32;   // BN = 32 / 96: the threads past the last row reload a valid row (unused)
          db[ks * BROWS + r] = ldg16(wbase + ((int64_t)(brow * p.wtaps + wt) * p.cin_pad + c0) * ES);
        }
      }
      if (++i_kc == kchunks) { i_kc = 0; ++i_tap; }
    };
    auto store_lds = [&](int buf, const uint4* sa, const uint4* sb) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int r = 0; r < MF; ++r)
          *reinterpret_cast<uint4*>(sA + buf * A_BYTES + ks * A_PLANE + lds_off2(lrow + r * 64, lseg)) = sa[ks * MF + r];
#pragma unroll
        for (int r = 0; r < BROWS; ++r) {
          const int brow = lrow + r * 64;
          if (BN % 64 == 0 || brow < BN)
            *reinterpret_cast<uint4*>(sB + buf * B_BYTES + ks * B_PLANE + lds_off2(brow, lseg)) = sb[ks * BROWS + r];
        }
      }
    };

#pragma unroll
    for (int j = 0; j < RING - 1; ++j) issue_loads(ra[j], rb[j], j < nsteps);
    store_lds(0, ra[0], rb[0]);
    __syncthreads();

    for (int s0 = 0; s0 < nsteps; s0 += RING) {
#pragma unroll
      for (int j = 0; j < RING; ++j) {
        const int s = s0 + j;
        if (s < nsteps) {                               // uniform
          const int buf = s & 1;
          issue_loads(ra[(j + RING - 1) % RING], rb[(j + RING - 1) % RING], s + RING - 1 < nsteps);
          const int frow = lane & 15, fseg = lane >> 4;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            uint4 fa[MF], fb[NF];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
              fa[mf] = *reinterpret_cast<const uint4*>(sA + buf * A_BYTES + ks * A_PLANE + lds_off2(wave * (16 * MF) + mf * 16 + frow, fseg));
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
              fb[nf] = *reinterpret_cast<const uint4*>(sB + buf * B_BYTES + ks * B_PLANE + lds_off2(nf * 16 + frow, fseg));
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
              for (int nf = 0; nf < NF; ++nf) Mma<T>::run(fa[mf], fb[nf], acc[mf][nf]);
          }
          if (s + 1 < nsteps) store_lds(buf ^ 1, ra[(j + 1) % RING], rb[(j + 1) % RING]);
          __syncthreads();
        }
      }
    }

    // ---- epilogue ----
    // acc[mf][nf][r] = D[row = wave*16*MF+mf*16+4*(lane>>4)+r][col = nf*16+(lane&15)]
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v0 = acc[mf][nf][r];
          if (!BNS) { st_s[nf] += v0; st_q[nf] += v0 * v0; }
          float v = v0 * e_scale[nf] + e_shift[nf];
          v = act_f(v, p.act);
          const int row = wave * (16 * MF) + mf * 16 + 4 * (lane >> 4) + r;
          const int col = nf * 16 + (lane & 15);
          *reinterpret_cast<T*>(sC + row * CROW + col * ES) = (T)v;
        }
      }
    __syncthreads();
    constexpr int VPR = BN / SEG;   // 16-byte vectors per row
    for (int v = tid; v < BM * VPR; v += THREADS) {
      const int row = v / VPR, cs = v - row * VPR;
      const int m = m0 + row;
      const int c0 = tn * BN + cs * SEG;
      if (m >= p.M || c0 >= p.Cout) continue;
      const int n = m / HWo; const int rem = m - n * HWo; const int oy = rem / p.Wo; const int ox = rem - oy * p.Wo;
      uint4 cv = *reinterpret_cast<const uint4*>(sC + row * CROW + cs * 16);
      float f[SEG];
      Vec<T>::unpack(cv, f);
      const int nvalid = (p.Cout - c0) < SEG ? (p.Cout - c0) : SEG;
      if (p.det_no > 0) {
        // Detect layout: [n, a, oy, ox, o] with conv channel = a*no + o
        T* yb = reinterpret_cast<T*>(p.y);
        const int na = p.Cout / p.det_no;
        for (int i = 0; i < nvalid; ++i) {
          const int c = c0 + i; const int a = c / p.det_no, o = c - a * p.det_no;
          yb[(((int64_t)(n * na + a) * p.Ho + oy) * p.Wo + ox) * p.det_no + o] = (T)f[i];
        }
        continue;
      }
      char* yp = p.y + ((int64_t)n * p.y_sn + (int64_t)oy * p.y_sh + (int64_t)ox * p.y_sw + c0) * ES;
      if (nvalid == SEG) {
        if (p.res) {
          float g[SEG];
          Vec<T>::unpack(ldg16(p.res + ((int64_t)n * p.r_sn + (int64_t)oy * p.r_sh + (int64_t)ox * p.r_sw + c0) * ES), g);
#pragma unroll
          for (int i = 0; i < SEG; ++i) f[i] += g[i];
        }
        if (p.accumulate) {
          float g[SEG];
          Vec<T>::unpack(ldg16(yp), g);
#pragma unroll
          for (int i = 0; i < SEG; ++i) f[i] += g[i];
        }
        const uint4 packed = Vec<T>::pack(f);
        stg16(yp, packed);
        if (BNS) {
          if (sgi >= 0) {
            const BnbSeg& sg = p.bnb.seg[sgi];
            if (c0 < sg.c1) {
              float gq[SEG], yv[SEG];
              Vec<T>::unpack(packed, gq);            // gout as stored (what the apply pass will read back)
              Vec<T>::unpack(ldg16(sg.y + ((int64_t)n * sg.y_sn + (int64_t)oy * sg.y_sh + (int64_t)ox * sg.y_sw + (c0 - sg.c0)) * ES), yv);
              const int cl = cs * SEG;
#pragma unroll
              for (int i = 0; i < SEG; ++i) {
                const float dz = gq[i] * act_grad_f(fmaf(yv[i], sBN[2 * BN + cl + i], sBN[3 * BN + cl + i]), sg.act);
                bs0[i] += dz;
                bs1[i] += dz * (yv[i] - sBN[cl + i]) * sBN[BN + cl + i];
              }
            }
          }
        }
      } else {
        T* ys = reinterpret_cast<T*>(yp);
        const T* rs = p.res ? reinterpret_cast<const T*>(p.res + ((int64_t)n * p.r_sn + (int64_t)oy * p.r_sh + (int64_t)ox * p.r_sw + c0) * ES) : nullptr;
        for (int i = 0; i < nvalid; ++i) {
          float v2 = f[i];
          if (rs) v2 += (float)rs[i];
          if (p.accumulate) v2 += (float)ys[i];
          ys[i] = (T)v2;
        }
      }
    }
    __syncthreads();
  }

  if (BNS) {
    if (sgi >= 0) {
      // threads with the same channel vector (tid % VPR) meet in LDS; one atomic per channel and sum per workgroup
      constexpr int VPRc = BN / SEG;
      float* red = reinterpret_cast<float*>(smem);     // [THREADS / VPR][2][BN]   (the last tile's epilogue ended with a barrier)
      const int cs = tid % VPRc, rr = tid / VPRc;
#pragma unroll
      for (int i = 0; i < SEG; ++i) { red[(rr * 2) * BN + cs * SEG + i] = bs0[i]; red[(rr * 2 + 1) * BN + cs * SEG + i] = bs1[i]; }
      __syncthreads();
      const BnbSeg& sg = p.bnb.seg[sgi];
      const int Cs = sg.c1 - sg.c0;
      for (int t = tid; t < 2 * BN; t += THREADS) {
        const int which = t / BN, cl = t - which * BN;
        float a = 0.f;
        for (int q = 0; q < THREADS / VPRc; ++q) a += red[(q * 2 + which) * BN + cl];
        const int ci = tn * BN + cl - sg.c0;
        if (ci < Cs) atomicAdd(sg.dsum + (blockIdx.x % MYOLO_STAT_COPIES) * 2 * Cs + which * Cs + ci, a);
      }
    }
  } else if (p.stats) {
    // lanes -> wave (rows of the fragment) -> workgroup (LDS) -> one coalesced atomic per channel per workgroup
    float* red = reinterpret_cast<float*>(smem);       // [4 waves][2*BN]; the last tile's epilogue ended with a barrier
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      float s = st_s[nf], q = st_q[nf];
      s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
      if (lane < 16) { red[wave * 2 * BN + nf * 16 + lane] = s; red[wave * 2 * BN + BN + nf * 16 + lane] = q; }
    }
    __syncthreads();
    for (int t = tid; t < 2 * BN; t += THREADS) {
      const float a = red[t] + red[2 * BN + t] + red[4 * BN + t] + red[6 * BN + t];
      const int cl = t < BN ? t : t - BN;
      const int c = tn * BN + cl;
      if (c < p.Cout) atomicAdd(p.stats + (blockIdx.x % MYOLO_STAT_COPIES) * 2 * p.Cout + (t < BN ? c : p.Cout + c), a);
    }
  }
}

template <typename T, int BN, int BM, int KS, int BNS>
int launch_conv4(ConvK k, int grid_x, int ntile_n, hipStream_t st) {
  constexpr int ES = (int)sizeof(T);
  constexpr int SEG = 16 / ES;
  constexpr int AB = 2 * KS * (BM * 64 + BN * 64);
  constexpr int CB = BM * (BN * ES + 16);
  constexpr int SB = 4 * 2 * BN * 4;             // statistics reduction [4 waves][2*BN] floats
  constexpr int RB = BNS ? (THREADS / (BN / SEG)) * 2 * BN * 4 : 0;   // BNS reduction [threads per channel vector][2][BN]
  int smem = (AB > CB ? AB : CB) > SB ? (AB > CB ? AB : CB) : SB;
  if (RB > smem) smem = RB;
  k.bn_off = smem;
  if (BNS) smem += 4 * BN * 4;
  auto kern = conv_igemm_kernel<T, BN, BM, KS, BNS>;
  MYOLO_ENSURE_DYN_SMEM(kern, smem);
  hipLaunchKernelGGL(kern, dim3(grid_x, ntile_n), dim3(THREADS), smem, st, k);
  MYOLO_CHECK_LAUNCH();
  return 0;
}
template <typename T, int BN, int BM, int KS>
int launch_conv3(const ConvK& k, int grid_x, int ntile_n, hipStream_t st) {
  return k.bnb.n > 0 ? launch_conv4<T, BN, BM, KS, 1>(k, grid_x, ntile_n, st) : launch_conv4<T, BN, BM, KS, 0>(k, grid_x, ntile_n, st);
}

template <typename T, int BN, int BM>
int launch_conv2(const ConvK& k, int grid_x, int ntile_n, hipStream_t st) {
  constexpr int KC = ET<T>::KC;
  // 128-byte K steps for the small-map tiles only: with BM = 128 the extra ring registers cost a wave per SIMD
  const bool wide = BM == 64 && (k.cin_pad % (2 * KC)) == 0 && k.ntaps * (k.cin_pad / (2 * KC)) >= 4;
  return wide ? launch_conv3<T, BN, BM, 2>(k, grid_x, ntile_n, st) : launch_conv3<T, BN, BM, 1>(k, grid_x, ntile_n, st);
}
template <typename T, int BN>
int launch_conv(const ConvK& k, int grid_x, int ntile_n, hipStream_t st) {
  return k.bm == 64 ? launch_conv2<T, BN, 64>(k, grid_x, ntile_n, st) : launch_conv2<T, BN, 128>(k, grid_x, ntile_n, st);
}

}  // namespace

static int g_ig_bm = 0, g_ig_bn = 0, g_ig_wgs = 0;      // 0: the rules below
int myolo_conv_igemm_set(const char* name, int value) {
  if (!strcmp(name, "igemm_bm")) { g_ig_bm = value; return 0; }
  if (!strcmp(name, "igemm_bn")) { g_ig_bn = value; return 0; }
  if (!strcmp(name, "igemm_wgs")) { g_ig_wgs = value; return 0; }
  return MYOLO_EINVAL;
}

extern "C" int myolo_conv(const myolo_conv_desc* d, void* stream) {
  if (!d || !d->x.ptr || !d->y.ptr || !d->w) return MYOLO_EINVAL;
  const int dt = d->x.dtype;
  if (dt != d->y.dtype || (dt != MYOLO_F16 && dt != MYOLO_F32)) return MYOLO_EINVAL;
  const int es = dt == MYOLO_F16 ? 2 : 4, seg = 16 / es, kc = 64 / es;
  if (d->ntaps < 1 || d->ntaps > MYOLO_MAX_TAPS) return MYOLO_EINVAL;
  if (d->cin_pad % kc || d->cout_pad % 32 || d->x.c > d->cin_pad || d->y.c > d->cout_pad) return MYOLO_EINVAL;
  if (d->x.c % seg || d->x.sw % seg || d->x.sh % seg || d->x.sn % seg || ((uintptr_t)d->x.ptr & 15)) return MYOLO_EINVAL;
  if (d->det_no > 0 && (d->y.c % d->det_no)) return MYOLO_EINVAL;
  if (d->res.ptr && d->res.dtype != dt) return MYOLO_EINVAL;
  const bool want_bnb = d->bnb != nullptr && d->nbnb > 0;
  if (want_bnb) {
    if (d->nbnb > MYOLO_MAX_BNB || d->det_no > 0) return MYOLO_EINVAL;
    for (int i = 0; i < d->nbnb; ++i) {
      const myolo_bn_bwd_seg& sg = d->bnb[i];
      if (sg.c0 < 0 || sg.c1 <= sg.c0 || sg.c1 > d->y.c || sg.c0 % seg || sg.c1 % seg || !sg.y.ptr || !sg.saved || !sg.gamma || !sg.beta ||
          !sg.dsum || sg.y.dtype != dt || sg.y.c != sg.c1 - sg.c0 || sg.y.n != d->y.n || sg.y.h != d->y.h || sg.y.w != d->y.w)
        return MYOLO_EINVAL;
    }
  }
  if (dt == MYOLO_F16) {
    int done = 0;
    int r = myolo_conv_small_try(d, stream);          // small maps, eval epilogue: split-K over the waves, operands straight from L2 (conv_small.hip)
    if (r != -1) return r;
    const int mid = myolo_conv_mid_mode();
    r = myolo_conv_midx_try(d, stream);                               // k x k stride-1 training layers: input halo resident in LDS, weights streamed (conv_midx.hip; MYOLO_CONV_MIDX)
    if (r == -1) r = mid >= 2 ? myolo_conv_mid_try(d, stream, &done) : -1;         // (mode 2: ahead of the halo / streaming kernels)
    if (r == -1) r = myolo_conv_halo_try(d, stream, &done);    // k x k stride-1 layers: input halo tiles staged in LDS (conv_halo.hip)
    if (r == -1) r = myolo_conv_stream_try(d, stream, &done);      // HBM-bound 1x1 / strided layers: the streaming kernel (conv_stream.hip)
    if (r == -1 && mid == 1) {                                      // mid-size / small training maps: LDS-DMA staged 8-wave tiles (conv_mid.hip)
      r = myolo_conv_mid_try(d, stream, &done);                     // (resets `done`: a *_try that declines late may have claimed the fold)
    }
    if (r != -1) {
      if (r) return r;
      return (want_bnb && !done) ? myolo_bnb_fallback(d, &d->y, stream) : 0;
    }
  }
  ConvK k;
  k.x = (const char*)d->x.ptr; k.x_sn = d->x.sn; k.x_sh = d->x.sh; k.x_sw = d->x.sw;
  k.Hi = d->x.h; k.Wi = d->x.w; k.Cin = d->x.c;
  k.y = (char*)d->y.ptr; k.y_sn = d->y.sn; k.y_sh = d->y.sh; k.y_sw = d->y.sw;
  k.Ho = d->y.h; k.Wo = d->y.w; k.Cout = d->y.c; k.N = d->y.n;
  k.w = (const char*)d->w; k.cin_pad = d->cin_pad; k.cout_pad = d->cout_pad; k.wtaps = d->wtaps;
  k.ntaps = d->ntaps; k.stride = d->stride; k.up = d->up_shift;
  for (int i = 0; i < MYOLO_MAX_TAPS; ++i) { k.tap_dy[i] = d->tap_dy[i]; k.tap_dx[i] = d->tap_dx[i]; k.tap_w[i] = d->tap_w[i]; }
  k.scale = d->scale; k.shift = d->shift; k.act = d->act; k.accumulate = d->accumulate;
  k.res = (const char*)d->res.ptr; k.r_sn = d->res.sn; k.r_sh = d->res.sh; k.r_sw = d->res.sw;
  k.stats = d->stats; k.det_no = d->det_no;
  k.bnb.n = 0; k.bn_off = 0;
  const int64_t M = (int64_t)k.N * k.Ho * k.Wo;
  if (M <= 0 || M > 0x7fffffff) return MYOLO_EINVAL;
  k.M = (int)M;
  // Tile shape from the number of tiles against the 768-1024 workgroups the chip holds at once (r3 tile sweep, batch-16 layers):
  //   * <= 256 tiles of 64 x 128 (one per CU or fewer): split N, 64 x 64 tiles (256 -> 256 at 8192 pixels: 11.6 -> 9.9 us);
  //   * 769 .. 1536 tiles of 64 rows would run a second, one-third-full round: 128-row tiles, one round (256 -> 256 at 32768 pixels:
  //     26.7 -> 22.4 us; 512 -> 256: 33.3 -> 27.5 us);
  //   * in between 64-row tiles (twice the workgroups of the 128-row ones on small maps), beyond it 128-row tiles.
  int bn = (d->cout_pad % 128 == 0) ? 128 : ((d->cout_pad % 64 == 0) ? 64 : 32);
  // 96-wide N tile (fp16): the 96-channel layers of yolov5m are ONE N tile instead of three 32-wide ones that each re-stage the input
  constexpr int no96 = 0;
  if (bn == 32 && d->cout_pad % 96 == 0 && dt == MYOLO_F16 && !no96) bn = 96;
  const int64_t mt64 = (M + 63) / 64;
  if (bn == 128 && mt64 * (d->cout_pad / 128) <= 256) bn = 64;
  if (g_ig_bn && bn > g_ig_bn && d->cout_pad % g_ig_bn == 0) bn = g_ig_bn;
  const int ntile_n = d->cout_pad / bn;
  k.bm = mt64 * ntile_n > 768 ? 128 : 64;
  if (g_ig_bm) k.bm = g_ig_bm;
  k.ntile_m = (int)((M + k.bm - 1) / k.bm);
  k.tiles_per_xcd = (k.ntile_m + 7) / 8;
  // persistent grid: ~3 workgroups per CU in total, multiple of 8 (one slot range per XCD)
  int per_xcd = ((g_ig_wgs ? g_ig_wgs : 768) / ntile_n + 7) / 8;
  if (per_xcd < 1) per_xcd = 1;
  if (per_xcd > k.tiles_per_xcd) per_xcd = k.tiles_per_xcd;
  const int grid_x = per_xcd * 8;
  hipStream_t st = (hipStream_t)stream;
  // (the folded BatchNorm-backward sums need a fixed channel vector per thread: 256 % (BN/8) == 0, not the 96-wide tile)
  const bool fold = want_bnb && !d->stats && !d->scale && !d->shift && d->act == MYOLO_ACT_NONE && bn != 96 && bnb_aligned(d, bn);
  if (fold) bnb_fill(&k.bnb, d);
  int r;
  if (dt == MYOLO_F16) {
    if (bn == 128) r = launch_conv<half_t, 128>(k, grid_x, ntile_n, st);
    else if (bn == 96) r = launch_conv<half_t, 96>(k, grid_x, ntile_n, st);
    else if (bn == 64) r = launch_conv<half_t, 64>(k, grid_x, ntile_n, st);
    else r = launch_conv<half_t, 32>(k, grid_x, ntile_n, st);
  } else {
    if (bn == 128) r = launch_conv<float, 128>(k, grid_x, ntile_n, st);
    else if (bn == 64) r = launch_conv<float, 64>(k, grid_x, ntile_n, st);
    else r = launch_conv<float, 32>(k, grid_x, ntile_n, st);
  }
  if (r) return r;
  return (want_bnb && !fold) ? myolo_bnb_fallback(d, &d->y, stream) : 0;
}
