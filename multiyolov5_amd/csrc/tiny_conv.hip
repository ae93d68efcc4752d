// 1x1 Conv2d (+ train-mode BatchNorm2d) (+ activation) over maps of at most MYOLO_TINY_MAX_PIX pixels: ONE workgroup per layer, up to
// MYOLO_TINY_MAX_GROUP independent layers per launch, forward and backward (BatchNorm-backward sums + apply + dgrad).
//
// Replaces, for PyramidPooling's four branch convolutions on the 1x1 / 2x2 / 3x3 / 6x6 pooled maps (reference models/common.py:521-537:
// `Conv(in_channels, in_channels // 4, k=1)` = nn.Conv2d + nn.BatchNorm2d + nn.SiLU, common.py:34-46) and any other Conv of that size:
//   forward   myolo_conv (raw output + statistics) + myolo_bn_act_fwd                      2 launches per layer -> 1 per GROUP
//   backward  myolo_bn_act_bwd_reduce + myolo_bn_act_bwd_apply + myolo_conv (dgrad)        3 launches per layer -> 1 per GROUP
// (the weight gradient stays myolo_conv_wgrad over the dy this kernel writes).  On these maps every launch is fixed latency: in the
// round-4 trace the eight forward launches of the four branches take 54 us and the twelve backward ones 166 us (the BatchNorm passes
// of a 16 x 2 x 2 x 32 tensor: 33 + 18 us beside the weight-gradient queue) for 4.7 MFLOP of work.
//
// One workgroup of 8 waves holds a whole layer, so the batch statistics need no second launch: the weights go to LDS once (fp32 master
// -> the plan dtype, = autocast's cast), a wave takes 16-pixel row blocks, the x fragments come straight from global memory in MFMA
// operand layout (a lane's 8 consecutive input channels of one pixel = one 16-byte load), D^T = W . X^T gives a lane four consecutive
// output channels of one pixel (8-byte stores), per-channel sums by DPP row sums + LDS atomics; after the barrier every lane re-reads
// its own raw values and writes the activation.  fp32 plans (parity mode) take scalar loops over the same LDS tables.
#include "myolo_dev.h"
#include <string.h>

namespace {

constexpr int TNT = 512;
constexpr int KSMAX = 16;            // 32-channel K steps of the forward (cin <= 512)

struct TinyArgs { myolo_tiny_conv_desc d[MYOLO_TINY_MAX_GROUP]; };

__device__ __forceinline__ float row_sum16(float v) {            // sum over the 16 lanes of a DPP row, valid in lane 15 of the row
#define TINY_SHR(n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + (n), 0xf, 0xf, true))
  v += TINY_SHR(1); v += TINY_SHR(2); v += TINY_SHR(4); v += TINY_SHR(8);
#undef TINY_SHR
  return v;
}

struct PixOf {                       // linear pixel -> element offset of a view
  int hw, w; int64_t sn, sh, sw;
  __device__ PixOf(const myolo_tensor& t) : hw(t.h * t.w), w(t.w), sn(t.sn), sh(t.sh), sw(t.sw) {}
  __device__ int64_t operator()(int p) const {
    const int n = p / hw, r = p - n * hw, y = r / w, x = r - y * w;
    return (int64_t)n * sn + (int64_t)y * sh + (int64_t)x * sw;
  }
};

// LDS: [2*Cout] sums | [4*Cout] per-channel constants | weights
__host__ __device__ constexpr int tab_floats(int Cout) { return 6 * Cout; }

// ------------------------------------------------------------------------------------------------ forward
template <typename T>
__global__ __launch_bounds__(TNT) void tiny_fwd_kernel(const TinyArgs a) {
  const myolo_tiny_conv_desc& d = a.d[blockIdx.x];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = d.x.n * d.x.h * d.x.w, Cin = d.x.c, Cout = d.z.c;
  float* ssum = reinterpret_cast<float*>(smem);                   // [2][Cout] sum, sum of squares
  float* tab = ssum + 2 * Cout;                                   // [2][Cout] scale, shift
  char* wl = smem + tab_floats(Cout) * 4;
  const bool bn = d.gamma != nullptr;
  const PixOf px(d.x), pz(d.z), po(d.out);
  const T* xb = reinterpret_cast<const T*>(d.x.ptr);
  T* zb = reinterpret_cast<T*>(d.z.ptr);
  T* ob = reinterpret_cast<T*>(d.out.ptr);
  for (int c = tid; c < 2 * Cout; c += TNT) ssum[c] = 0.f;

  if constexpr (sizeof(T) == 2) {
    // ---- fp16: MFMA.  weights [Cout][Cin + 8] halves (an odd number of 16-byte units per row for Cin % 16 == 0)
    const int pitch = Cin + 8;
    half_t* W = reinterpret_cast<half_t*>(wl);
    for (int i = tid * 4; i < Cout * Cin; i += TNT * 4) {
      const float4 v = *reinterpret_cast<const float4*>(d.w + i);
      const int r = i / Cin, c = i - r * Cin;
      h4_t h; h[0] = (half_t)v.x; h[1] = (half_t)v.y; h[2] = (half_t)v.z; h[3] = (half_t)v.w;
      *reinterpret_cast<h4_t*>(W + r * pitch + c) = h;
    }
    __syncthreads();
    const int lq = lane >> 4, l15 = lane & 15;
    const int nrb = (P + 15) >> 4, KS = Cin >> 5, CB = Cout >> 4;
    auto load_x = [&](int rb, uint4* f) {                         // the row block's x fragments: pixel rb*16 + l15, channels ks*32 + lq*8 ..
      const int p = rb * 16 + l15;
      const bool ok = p < P;
      const T* src = xb + (ok ? px(p) : 0) + lq * 8;
#pragma unroll
      for (int ks = 0; ks < KSMAX; ++ks)
        if (ks < KS) f[ks] = ok ? ldg16(src + ks * 32) : uint4{0u, 0u, 0u, 0u};
    };
    uint4 cur[KSMAX], nxt[KSMAX];
#pragma unroll
    for (int ks = 0; ks < KSMAX; ++ks) { cur[ks] = uint4{0u, 0u, 0u, 0u}; nxt[ks] = cur[ks]; }
    if (wave < nrb) load_x(wave, cur);
    for (int rb = wave; rb < nrb; rb += 8) {
      if (rb + 8 < nrb) load_x(rb + 8, nxt);
      const int p = rb * 16 + l15;
      const bool ok = p < P;
      T* zp = zb + (ok ? pz(p) : 0);
      for (int cb = 0; cb < CB; ++cb) {
        f4_t acc = f4_t{0.f, 0.f, 0.f, 0.f};
        const half_t* wr = W + (cb * 16 + l15) * pitch + lq * 8;
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks)
          if (ks < KS) {
            const h8_t af = *reinterpret_cast<const h8_t*>(wr + ks * 32);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, *reinterpret_cast<const h8_t*>(&cur[ks]), acc, 0, 0, 0);
          }
        // acc[r] = z[pixel rb*16 + l15][channel cb*16 + 4*lq + r]
        const int co = cb * 16 + 4 * lq;
        if (bn) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = ok ? acc[r] : 0.f;
            const float s = row_sum16(v), q = row_sum16(v * v);
            if (l15 == 15) { atomicAdd(&ssum[co + r], s); atomicAdd(&ssum[Cout + co + r], q); }
          }
        }
        if (ok) {
          h4_t h; h[0] = (half_t)acc[0]; h[1] = (half_t)acc[1]; h[2] = (half_t)acc[2]; h[3] = (half_t)acc[3];
          *reinterpret_cast<h4_t*>(zp + co) = h;
        }
      }
#pragma unroll
      for (int ks = 0; ks < KSMAX; ++ks) cur[ks] = nxt[ks];
    }
  } else {
    // ---- fp32 (parity mode): scalar loops, weights [Cout][Cin] floats
    float* W = reinterpret_cast<float*>(wl);
    for (int i = tid; i < Cout * Cin; i += TNT) W[i] = d.w[i];
    __syncthreads();
    for (int i = tid; i < P * Cout; i += TNT) {
      const int p = i / Cout, c = i - p * Cout;
      const T* xr = xb + px(p);
      const float* wr = W + c * Cin;
      float acc = 0.f;
      for (int k = 0; k < Cin; ++k) acc = fmaf((float)xr[k], wr[k], acc);
      zb[pz(p) + c] = (T)acc;
      if (bn) { atomicAdd(&ssum[c], acc); atomicAdd(&ssum[Cout + c], acc * acc); }
    }
  }
  __syncthreads();
  for (int c = tid; c < Cout; c += TNT) {
    float sc = 1.f, sh = 0.f;
    if (bn) {
      const double mean = (double)ssum[c] / (double)P;
      double var = (double)ssum[Cout + c] / (double)P - mean * mean;
      if (var < 0.0) var = 0.0;
      const float invstd = rsqrtf((float)var + d.eps);
      sc = d.gamma[c] * invstd;
      sh = d.beta[c] - (float)mean * sc;
      if (d.saved) { d.saved[c] = (float)mean; d.saved[Cout + c] = invstd; }
      if (d.running_mean) {
        d.running_mean[c] = (1.f - d.momentum) * d.running_mean[c] + d.momentum * (float)mean;
        const float unb = P > 1 ? (float)var * (float)P / (float)(P - 1) : (float)var;
        d.running_var[c] = (1.f - d.momentum) * d.running_var[c] + d.momentum * unb;
      }
    }
    tab[c] = sc; tab[Cout + c] = sh;
  }
  if (tid == 0 && bn && d.nbt) *d.nbt += 1;
  __syncthreads();
  // ---- second pass: every thread re-reads the raw values IT stored (program order: no fence needed) and writes the activation
  if constexpr (sizeof(T) == 2) {
    const int lq = lane >> 4, l15 = lane & 15;
    const int nrb = (P + 15) >> 4, CB = Cout >> 4;
    for (int rb = wave; rb < nrb; rb += 8) {
      const int p = rb * 16 + l15;
      if (p >= P) continue;
      const T* zp = zb + pz(p);
      T* op = ob + po(p);
      for (int cb = 0; cb < CB; ++cb) {
        const int co = cb * 16 + 4 * lq;
        const h4_t zv = *reinterpret_cast<const h4_t*>(zp + co);
        h4_t o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (half_t)act_f(fmaf((float)zv[r], tab[co + r], tab[Cout + co + r]), d.act);
        *reinterpret_cast<h4_t*>(op + co) = o;
      }
    }
  } else {
    for (int i = tid; i < P * Cout; i += TNT) {
      const int p = i / Cout, c = i - p * Cout;
      ob[po(p) + c] = (T)act_f(fmaf((float)zb[pz(p) + c], tab[c], tab[Cout + c]), d.act);
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
// dz = gout * act'(bn(z));  dy = sc*dz + cb*z + cd  (bn_act.hip's folded form: cb = -sc*k1*invstd, cd = -sc*k0 - cb*mean, k = sums / P);
// dgamma += sum dz*xhat, dbeta += sum dz;  gx (+)= dy . W
template <typename T>
__global__ __launch_bounds__(TNT) void tiny_bwd_kernel(const TinyArgs a) {
  const myolo_tiny_conv_desc& d = a.d[blockIdx.x];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int SEG = ET<T>::SEG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = d.x.n * d.x.h * d.x.w, Cin = d.x.c, Cout = d.z.c;
  float* ssum = reinterpret_cast<float*>(smem);                   // [2][Cout]: sum dz, sum dz*xhat
  float* tab = ssum + 2 * Cout;                                   // [4][Cout]: sc, sh, cb, cd
  char* wl = smem + tab_floats(Cout) * 4;
  const bool bn = d.gamma != nullptr;
  const PixOf pz(d.z), pg(d.gout), pd(d.dy), px(d.gx.ptr ? d.gx : d.x);
  const T* zb = reinterpret_cast<const T*>(d.z.ptr);
  const T* gb = reinterpret_cast<const T*>(d.gout.ptr);
  T* db = reinterpret_cast<T*>(d.dy.ptr);
  T* xgb = reinterpret_cast<T*>(d.gx.ptr);
  for (int c = tid; c < 2 * Cout; c += TNT) ssum[c] = 0.f;
  for (int c = tid; c < Cout; c += TNT) {
    float sc = 1.f, sh = 0.f;
    if (bn) { sc = d.gamma[c] * d.saved[Cout + c]; sh = d.beta[c] - d.saved[c] * sc; }
    tab[c] = sc; tab[Cout + c] = sh; tab[2 * Cout + c] = 0.f; tab[3 * Cout + c] = 0.f;
  }
  // weights for the dgrad: fp16 W^T [Cin][KP + 8] halves (KP = Cout rounded up to 32, the pad columns ZERO: they meet zero dy
  // fragments, and 0 * garbage could be NaN); fp32: [Cout][Cin] floats
  const int KP = (Cout + 31) & ~31, pitchT = KP + 8;
  if (d.gx.ptr) {
    if constexpr (sizeof(T) == 2) {
      half_t* WT = reinterpret_cast<half_t*>(wl);
      for (int i = tid; i < Cin * pitchT; i += TNT) WT[i] = (half_t)0.f;
      __syncthreads();
      for (int i = tid; i < Cout * Cin; i += TNT) {
        const int co = i / Cin, ci = i - co * Cin;
        WT[ci * pitchT + co] = (half_t)d.w[i];
      }
    } else {
      float* W = reinterpret_cast<float*>(wl);
      for (int i = tid; i < Cout * Cin; i += TNT) W[i] = d.w[i];
    }
  }
  __syncthreads();
  // thread layout of the elementwise passes: channel group cg (SEG channels) is fixed per thread, pixels stride by PPB
  const int G = Cout / SEG, PPB = TNT / G;
  const int cg = tid % G, pl = tid / G;
  const bool act_thr = pl < PPB;                                  // (TNT % G != 0: the last partial row of threads idles)
  const int c0 = cg * SEG;
  if (bn) {
    float s0[SEG], s1[SEG], sc[SEG], sh[SEG], mean[SEG], istd[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) {
      s0[i] = 0.f; s1[i] = 0.f; sc[i] = tab[c0 + i]; sh[i] = tab[Cout + c0 + i]; mean[i] = d.saved[c0 + i]; istd[i] = d.saved[Cout + c0 + i];
    }
    if (act_thr)
      for (int p = pl; p < P; p += PPB) {
        float fz[SEG], fg[SEG];
        Vec<T>::unpack(ldg16(zb + pz(p) + c0), fz);
        Vec<T>::unpack(ldg16(gb + pg(p) + c0), fg);
#pragma unroll
        for (int i = 0; i < SEG; ++i) {
          const float dz = fg[i] * act_grad_f(fmaf(fz[i], sc[i], sh[i]), d.act);
          s0[i] += dz;
          s1[i] += dz * (fz[i] - mean[i]) * istd[i];
        }
      }
    if (act_thr) {
#pragma unroll
      for (int i = 0; i < SEG; ++i) { atomicAdd(&ssum[c0 + i], s0[i]); atomicAdd(&ssum[Cout + c0 + i], s1[i]); }
    }
    __syncthreads();
    for (int c = tid; c < Cout; c += TNT) {
      const float k0 = ssum[c] / (float)P, k1 = ssum[Cout + c] / (float)P;
      const float sc1 = tab[c], istd1 = d.saved[Cout + c], mean1 = d.saved[c];
      const float cb = -sc1 * k1 * istd1;
      tab[2 * Cout + c] = cb;
      tab[3 * Cout + c] = -sc1 * k0 - cb * mean1;
      if (d.dgamma) d.dgamma[c] += ssum[Cout + c];
      if (d.dbeta) d.dbeta[c] += ssum[c];
    }
    __syncthreads();
  }
  // dy, elementwise (16-byte vectors)
  if (act_thr) {
    float sc[SEG], sh[SEG], cb[SEG], cd[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) { sc[i] = tab[c0 + i]; sh[i] = tab[Cout + c0 + i]; cb[i] = tab[2 * Cout + c0 + i]; cd[i] = tab[3 * Cout + c0 + i]; }
    for (int p = pl; p < P; p += PPB) {
      float fz[SEG], fg[SEG], o[SEG];
      Vec<T>::unpack(ldg16(zb + pz(p) + c0), fz);
      Vec<T>::unpack(ldg16(gb + pg(p) + c0), fg);
#pragma unroll
      for (int i = 0; i < SEG; ++i) {
        const float dz = fg[i] * act_grad_f(fmaf(fz[i], sc[i], sh[i]), d.act);
        o[i] = fmaf(sc[i], dz, fmaf(cb[i], fz[i], cd[i]));
      }
      stg16(db + pd(p) + c0, Vec<T>::pack(o));
    }
  }
  if (!d.gx.ptr) return;
  __syncthreads();                                                // dy of the whole layer is visible to the workgroup
  if constexpr (sizeof(T) == 2) {
    const half_t* WT = reinterpret_cast<const half_t*>(wl);
    const int lq = lane >> 4, l15 = lane & 15;
    const int nrb = (P + 15) >> 4, KS2 = KP >> 5, CIB = Cin >> 4;
    for (int rb = wave; rb < nrb; rb += 8) {
      const int p = rb * 16 + l15;
      const bool ok = p < P;
      uint4 bf[4];                                                // dy fragments: pixel p, channels ks*32 + lq*8 .. (Cout <= 128)
      const T* dp = db + (ok ? pd(p) : 0) + lq * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        if (ks < KS2) bf[ks] = (ok && ks * 32 + lq * 8 < Cout) ? ldg16(dp + ks * 32) : uint4{0u, 0u, 0u, 0u};
      T* gp = xgb + (ok ? px(p) : 0);
      for (int cib = 0; cib < CIB; ++cib) {
        f4_t acc = f4_t{0.f, 0.f, 0.f, 0.f};
        const half_t* wr = WT + (cib * 16 + l15) * pitchT + lq * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          if (ks < KS2)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const h8_t*>(wr + ks * 32), *reinterpret_cast<const h8_t*>(&bf[ks]), acc, 0, 0, 0);
        // acc[r] = gx[pixel p][channel cib*16 + 4*lq + r]
        if (ok) {
          const int ci = cib * 16 + 4 * lq;
          h4_t o;
          if (d.gx_accumulate) {
            const h4_t old = *reinterpret_cast<const h4_t*>(gp + ci);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (half_t)((float)old[r] + acc[r]);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (half_t)acc[r];
          }
          *reinterpret_cast<h4_t*>(gp + ci) = o;
        }
      }
    }
  } else {
    const float* W = reinterpret_cast<const float*>(wl);
    for (int i = tid; i < P * Cin; i += TNT) {
      const int p = i / Cin, ci = i - p * Cin;
      const T* dr = db + pd(p);
      float acc = 0.f;
      for (int co = 0; co < Cout; ++co) acc = fmaf((float)dr[co], W[co * Cin + ci], acc);
      T* g = xgb + px(p) + ci;
      *g = (T)(d.gx_accumulate ? (float)*g + acc : acc);
    }
  }
}

// ------------------------------------------------------------------------------------------------ host
inline bool tv_ok(const myolo_tensor& t, int dtype, int seg) {
  return t.ptr && t.dtype == dtype && t.c % seg == 0 && t.sw % seg == 0 && t.sh % seg == 0 && t.sn % seg == 0 && ((uintptr_t)t.ptr & 15) == 0;
}
inline bool same_pix(const myolo_tensor& a, const myolo_tensor& b) { return a.n == b.n && a.h == b.h && a.w == b.w; }

constexpr int TINY_LDS_MAX = 150 * 1024;
// dynamic LDS bytes of a layer (per-channel tables + weights), -1 when its shape is outside what the kernels hold
int tiny_lds(int dtype, int64_t P, int Cin, int Cout, bool bwd, bool with_gx) {
  if (dtype != MYOLO_F16 && dtype != MYOLO_F32) return -1;
  if (P < 1 || P > MYOLO_TINY_MAX_PIX || Cin < 1 || Cin > 32 * KSMAX || Cout < 1 || Cout > 128) return -1;
  if (dtype == MYOLO_F16 ? (Cin % 32 || Cout % 16) : (Cin % 4 || Cout % 4)) return -1;
  int wbytes;
  if (!bwd) wbytes = dtype == MYOLO_F16 ? Cout * (Cin + 8) * 2 : Cout * Cin * 4;
  else wbytes = !with_gx ? 0 : (dtype == MYOLO_F16 ? Cin * (((Cout + 31) & ~31) + 8) * 2 : Cout * Cin * 4);
  const int smem = tab_floats(Cout) * 4 + wbytes;
  return smem <= TINY_LDS_MAX ? smem : -1;
}

// 0 = ok; *smem = dynamic LDS bytes of the layer
int tiny_check(const myolo_tiny_conv_desc& d, bool bwd, int dtype, int* smem) {
  if (dtype != MYOLO_F16 && dtype != MYOLO_F32) return MYOLO_EINVAL;
  const int seg = dtype == MYOLO_F16 ? 8 : 4;
  if (!tv_ok(d.x, dtype, seg) || !tv_ok(d.z, dtype, seg) || !d.w) return MYOLO_EINVAL;
  if (!same_pix(d.x, d.z)) return MYOLO_EINVAL;
  const int Cin = d.x.c, Cout = d.z.c;
  if (d.gamma && (!d.beta || !d.saved)) return MYOLO_EINVAL;
  if (d.act != MYOLO_ACT_NONE && d.act != MYOLO_ACT_SILU && d.act != MYOLO_ACT_SIGMOID) return MYOLO_EINVAL;
  if (!bwd) {
    if (!tv_ok(d.out, dtype, seg) || !same_pix(d.out, d.z) || d.out.c != Cout) return MYOLO_EINVAL;
  } else {
    if (!tv_ok(d.gout, dtype, seg) || !tv_ok(d.dy, dtype, seg) || !same_pix(d.gout, d.z) || !same_pix(d.dy, d.z) || d.gout.c != Cout || d.dy.c != Cout)
      return MYOLO_EINVAL;
    if (d.gx.ptr && (!tv_ok(d.gx, dtype, seg) || !same_pix(d.gx, d.x) || d.gx.c != Cin)) return MYOLO_EINVAL;
  }
  *smem = tiny_lds(dtype, (int64_t)d.x.n * d.x.h * d.x.w, Cin, Cout, bwd, d.gx.ptr != nullptr);
  return *smem >= 0 ? 0 : MYOLO_EINVAL;
}

template <typename T>
int tiny_launch(bool bwd, const TinyArgs& a, int n, int smem, hipStream_t st) {
  if (bwd) {
    auto k = tiny_bwd_kernel<T>;
    MYOLO_ENSURE_DYN_SMEM(k, smem);
    hipLaunchKernelGGL(k, dim3(n), dim3(TNT), smem, st, a);
  } else {
    auto k = tiny_fwd_kernel<T>;
    MYOLO_ENSURE_DYN_SMEM(k, smem);
    hipLaunchKernelGGL(k, dim3(n), dim3(TNT), smem, st, a);
  }
  MYOLO_CHECK_LAUNCH();
  return 0;
}

int tiny_go(const myolo_tiny_conv_desc* d, int n, bool bwd, void* stream) {
  if (!d || n < 1 || n > MYOLO_TINY_MAX_GROUP) return MYOLO_EINVAL;
  TinyArgs a;
  memset(&a, 0, sizeof(a));
  const int dtype = d[0].x.dtype;
  int smem = 0;
  for (int i = 0; i < n; ++i) {
    int s = 0;
    const int r = tiny_check(d[i], bwd, dtype, &s);
    if (r) return r;
    smem = s > smem ? s : smem;
    a.d[i] = d[i];
  }
  hipStream_t st = (hipStream_t)stream;
  return dtype == MYOLO_F16 ? tiny_launch<half_t>(bwd, a, n, smem, st) : tiny_launch<float>(bwd, a, n, smem, st);
}

}  // namespace

extern "C" int myolo_tiny_conv_ok(int dtype, int pixels, int cin, int cout) {
  return tiny_lds(dtype, pixels, cin, cout, false, true) >= 0 && tiny_lds(dtype, pixels, cin, cout, true, true) >= 0;
}
extern "C" int myolo_tiny_conv_fwd(const myolo_tiny_conv_desc* d, int n, void* stream) { return tiny_go(d, n, false, stream); }
extern "C" int myolo_tiny_conv_bwd(const myolo_tiny_conv_desc* d, int n, void* stream) { return tiny_go(d, n, true, stream); }
