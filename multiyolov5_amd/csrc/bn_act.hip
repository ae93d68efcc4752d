// Training-mode BatchNorm2d (+SiLU) (+Bottleneck residual) around a raw conv output, forward and backward.
// HBM-bound elementwise / reduction kernels: 16-byte NHWC vectors, fp32 math, per-block scale/shift table in LDS.
// The batch statistics themselves come for free from the conv kernel's epilogue (myolo_conv `stats`).
// Replaces nn.BatchNorm2d (batch-stat path) + nn.SiLU + `x + cv2(cv1(x))` (reference models/common.py:43,105;
// eps 1e-3 / momentum 0.03 from utils/torch_utils.py:150-151) and their autograd backward.
#include "myolo_dev.h"
#include <string.h>

namespace {

// second parameter set for channels >= cs (two BatchNorm modules normalising the halves of ONE merged convolution: C3's cv2 | cv1)
struct BnSplit {
  int cs;
  int world;     // ranks whose statistics the arrays hold (myolo_bn_split.count_scale; 1 = this tensor only)
  const float* gamma2; const float* beta2; float* rm2; float* rv2; int64_t* nbt2; float* dgamma2; float* dbeta2;
};

struct PixDec {  // linear pixel -> (n,y,x) of a view
  int hw, w;
  __device__ PixDec(const myolo_tensor& t) : hw(t.h * t.w), w(t.w) {}
  __device__ void get(int64_t pix, int& n, int& y, int& x) const {
    n = (int)(pix / hw);
    const int r = (int)(pix - (int64_t)n * hw);
    y = r / w;
    x = r - y * w;
  }
};

// all views of a plan slice CHANNELS of dense NHWC buffers: pixel p of such a view sits at ptr + p*sw (no div/mod per vector)
__device__ __forceinline__ bool pix_dense(const myolo_tensor& t) { return t.sh == (int64_t)t.w * t.sw && t.sn == (int64_t)t.h * t.sh; }
template <typename T>
__device__ __forceinline__ T* pptr(const myolo_tensor& t, bool dense, const PixDec& pd, int64_t pix) {
  if (dense) return reinterpret_cast<T*>(t.ptr) + pix * t.sw;
  int n, yy, xx;
  pd.get(pix, n, yy, xx);
  return vptr<T>(t, n, yy, xx);
}

// Thread layout shared by the three passes: a workgroup is G*PPB threads over a SLICE of CW = G*SEG channels (blockIdx.y; CW = C below
// 128 channels, else 64 = one 128-byte line per pixel) and PPB = 256/G pixels; thread (cg = tid % G, pl = tid / G) always owns channel
// group cg of the slice, so its per-channel constants live in registers and the pixel index advances by a fixed stride -- no division
// and no table read per 16-byte vector.  The slice bounds the per-workgroup prologue (2*MYOLO_STAT_COPIES partial sums per channel:
// 128 loads per thread at C = 512 unsliced, 12.0 us for an 8 MB tensor against 8.0 us at C = 128; r3 bn_ubench).
template <typename T>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(myolo_tensor y, const float* __restrict__ stats,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* rm, float* rv, int64_t* nbt, float* saved, float eps,
                                                         float mom, int act, myolo_tensor res, myolo_tensor out, int G, int PPB,
                                                         BnSplit sp) {
  constexpr int SEG = ET<T>::SEG;
  extern __shared__ float tab[];  // [2*CW]: scale, shift of the slice
  const int C = y.c, CW = G * SEG, c0 = blockIdx.y * CW;
  const int64_t M = (int64_t)y.n * y.h * y.w;
  for (int cl = threadIdx.x; cl < CW; cl += blockDim.x) {
    const int c = c0 + cl;
    float sc = 1.f, sh = 0.f;
    if (gamma) {
      // the copies are combined and E[x^2] - mean^2 is formed in fp64: the fp32 partial sums are exact to ~1e-7 each, the
      // cancellation (|mean| >> std channels) happens in double
      float sf[4] = {0.f, 0.f, 0.f, 0.f}, qf[4] = {0.f, 0.f, 0.f, 0.f};      // 4 independent chains: 64 loads in flight, short add chains
#pragma unroll
      for (int k = 0; k < MYOLO_STAT_COPIES; ++k) { sf[k & 3] += stats[k * 2 * C + c]; qf[k & 3] += stats[k * 2 * C + C + c]; }
      const double ssum = ((double)sf[0] + (double)sf[1]) + ((double)sf[2] + (double)sf[3]);
      const double qsum = ((double)qf[0] + (double)qf[1]) + ((double)qf[2] + (double)qf[3]);
      const int64_t Mg = M * sp.world;                         // samples behind the sums (SyncBatchNorm: all ranks)
      const double meand = ssum / (double)Mg;
      double vard = qsum / (double)Mg - meand * meand;
      const float mean = (float)meand;
      float var = vard > 0.0 ? (float)vard : 0.f;
      const float invstd = rsqrtf(var + eps);
      const bool lo = c < sp.cs;
      const int cc = lo ? c : c - sp.cs;
      sc = (lo ? gamma : sp.gamma2)[cc] * invstd;
      sh = (lo ? beta : sp.beta2)[cc] - mean * sc;
      if (blockIdx.x == 0) {
        if (saved) { saved[c] = mean; saved[C + c] = invstd; }
        float* rmp = lo ? rm : sp.rm2; float* rvp = lo ? rv : sp.rv2;
        if (rmp) {
          rmp[cc] = (1.f - mom) * rmp[cc] + mom * mean;
          const float unb = Mg > 1 ? var * (float)Mg / (float)(Mg - 1) : var;
          rvp[cc] = (1.f - mom) * rvp[cc] + mom * unb;
        }
      }
    }
    tab[cl] = sc;
    tab[CW + cl] = sh;
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && nbt && gamma) *nbt += 1;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 1 && sp.nbt2 && gamma && sp.cs < C) *sp.nbt2 += 1;
  __syncthreads();
  const int cg = threadIdx.x % G, pl = threadIdx.x / G;
  const int co = c0 + cg * SEG;          // first channel of this thread's 16-byte vector
  float sc[SEG], sh[SEG];
#pragma unroll
  for (int i = 0; i < SEG; ++i) { sc[i] = tab[cg * SEG + i]; sh[i] = tab[CW + cg * SEG + i]; }
  const PixDec pd(y);
  const bool dense = pix_dense(y) && pix_dense(out) && (!res.ptr || pix_dense(res));
  const int64_t stride = (int64_t)gridDim.x * PPB;
  for (int64_t pix = (int64_t)blockIdx.x * PPB + pl; pix < M; pix += 2 * stride) {      // two pixels in flight per thread
    const int64_t pix2 = pix + stride;
    const bool has2 = pix2 < M;
    const int64_t q = has2 ? pix2 : pix;
    const uint4 r1 = ldg16(pptr<T>(y, dense, pd, pix) + co), r2 = ldg16(pptr<T>(y, dense, pd, q) + co);
    uint4 a1 = uint4{0u, 0u, 0u, 0u}, a2 = a1;
    if (res.ptr) { a1 = ldg16(pptr<T>(res, dense, pd, pix) + co); a2 = ldg16(pptr<T>(res, dense, pd, q) + co); }
    float f[SEG], f2[SEG];
    Vec<T>::unpack(r1, f); Vec<T>::unpack(r2, f2);
#pragma unroll
    for (int i = 0; i < SEG; ++i) { f[i] = act_f(fmaf(f[i], sc[i], sh[i]), act); f2[i] = act_f(fmaf(f2[i], sc[i], sh[i]), act); }
    if (res.ptr) {
      float g[SEG], g2[SEG];
      Vec<T>::unpack(a1, g); Vec<T>::unpack(a2, g2);
#pragma unroll
      for (int i = 0; i < SEG; ++i) { f[i] += g[i]; f2[i] += g2[i]; }
    }
    stg16(pptr<T>(out, dense, pd, pix) + co, Vec<T>::pack(f));
    if (has2) stg16(pptr<T>(out, dense, pd, pix2) + co, Vec<T>::pack(f2));
  }
}

// z and xhat of one element from the saved (mean, invstd)
struct BnCoef { float sc, sh, mean, invstd; };

template <typename T>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(myolo_tensor gout, myolo_tensor y,
                                                                const float* __restrict__ saved,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, int act, float* dsum,
                                                                int G, int PPB, BnSplit sp) {
  constexpr int SEG = ET<T>::SEG;
  extern __shared__ float red[];  // [PPB][G*SEG*2]
  const int C = y.c, c0 = blockIdx.y * (G * SEG);
  const int cg = threadIdx.x % G, pl = threadIdx.x / G;
  const int co = c0 + cg * SEG;
  const int64_t M = (int64_t)y.n * y.h * y.w;
  float sc[SEG], sh[SEG], mean[SEG], istd[SEG], s0[SEG], s1[SEG];
#pragma unroll
  for (int i = 0; i < SEG; ++i) {
    const int c = co + i;
    mean[i] = saved[c]; istd[i] = saved[C + c];
    const bool lo = c < sp.cs;
    const int cc = lo ? c : c - sp.cs;
    sc[i] = (lo ? gamma : sp.gamma2)[cc] * istd[i]; sh[i] = (lo ? beta : sp.beta2)[cc] - mean[i] * sc[i];
    s0[i] = 0.f; s1[i] = 0.f;
  }
  const PixDec pd(y);
  const bool dense = pix_dense(y) && pix_dense(gout);
  const int64_t stride = (int64_t)gridDim.x * PPB;
  constexpr int NPF = 4;                 // pixels in flight per thread (8 x 16-byte loads): a thread of a small map makes 1-2 round trips
  for (int64_t pix = (int64_t)blockIdx.x * PPB + pl; pix < M; pix += NPF * stride) {
    uint4 ry[NPF], rg[NPF];
    float wk[NPF];
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const int64_t pk = pix + k * stride;
      const int64_t q = pk < M ? pk : pix;                    // clamped: the loads stay unconditional, the weight masks
      wk[k] = pk < M ? 1.f : 0.f;
      ry[k] = ldg16(pptr<T>(y, dense, pd, q) + co);
      rg[k] = ldg16(pptr<T>(gout, dense, pd, q) + co);
    }
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      float fy[SEG], fg[SEG];
      Vec<T>::unpack(ry[k], fy); Vec<T>::unpack(rg[k], fg);
#pragma unroll
      for (int i = 0; i < SEG; ++i) {
        const float dz = wk[k] * fg[i] * act_grad_f(fy[i] * sc[i] + sh[i], act);
        s0[i] += dz;
        s1[i] += dz * (fy[i] - mean[i]) * istd[i];
      }
    }
  }
  float* mine = red + (size_t)pl * (G * SEG * 2) + cg * SEG * 2;
#pragma unroll
  for (int i = 0; i < SEG; ++i) { mine[2 * i] = s0[i]; mine[2 * i + 1] = s1[i]; }
  __syncthreads();
  for (int j = threadIdx.x; j < G * SEG * 2; j += blockDim.x) {
    float a = 0.f;
    for (int q = 0; q < PPB; ++q) a += red[(size_t)q * (G * SEG * 2) + j];
    const int c = c0 + (j >> 1);
    atomicAdd(dsum + (blockIdx.x % MYOLO_STAT_COPIES) * 2 * C + ((j & 1) ? C + c : c), a);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(myolo_tensor gout, myolo_tensor y,
                                                               const float* __restrict__ saved,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int act,
                                                               const float* __restrict__ dsum, float* dgamma,
                                                               float* dbeta, myolo_tensor dy, myolo_tensor gres,
                                                               int gres_acc, int G, int PPB, BnSplit sp) {
  constexpr int SEG = ET<T>::SEG;
  // dx = sc*(dz - k0 - xhat*k1), dz = gout*act'(y*sc + sh), xhat = (y - mean)*invstd, k = dsum/M
  //    = sc*dz + cb*y + cd   with cb = -sc*k1*invstd, cd = -sc*k0 - cb*mean
  extern __shared__ float tab[];  // [4*CW]: sc, sh, cb, cd of the slice
  const int C = y.c, CW = G * SEG, c0 = blockIdx.y * CW;
  const int64_t M = (int64_t)y.n * y.h * y.w;
  for (int cl = threadIdx.x; cl < CW; cl += blockDim.x) {
    const int c = c0 + cl;
    if (gamma) {
      const float mean = saved[c], istd = saved[C + c];
      const bool lo = c < sp.cs;
      const int cc = lo ? c : c - sp.cs;
      const float sc = (lo ? gamma : sp.gamma2)[cc] * istd;
      float d0 = 0.f, d1 = 0.f;
#pragma unroll
      for (int k = 0; k < MYOLO_STAT_COPIES; ++k) { d0 += dsum[k * 2 * C + c]; d1 += dsum[k * 2 * C + C + c]; }
      const float Mg = (float)M * (float)sp.world;
      const float k0 = d0 / Mg, k1 = d1 / Mg;
      const float cb = -sc * k1 * istd;
      tab[cl] = sc; tab[CW + cl] = (lo ? beta : sp.beta2)[cc] - mean * sc; tab[2 * CW + cl] = cb; tab[3 * CW + cl] = -sc * k0 - cb * mean;
      if (blockIdx.x == 0) {
        float* dgp = lo ? dgamma : sp.dgamma2; float* dbp = lo ? dbeta : sp.dbeta2;
        const float rw = 1.f / (float)sp.world;              // (exact for the power-of-two worlds of one node; 1 without SyncBatchNorm)
        if (dgp) dgp[cc] += sp.world > 1 ? d1 * rw : d1;
        if (dbp) dbp[cc] += sp.world > 1 ? d0 * rw : d0;
      }
    } else {
      tab[cl] = 1.f; tab[CW + cl] = 0.f; tab[2 * CW + cl] = 0.f; tab[3 * CW + cl] = 0.f;
    }
  }
  __syncthreads();
  const int cg = threadIdx.x % G, pl = threadIdx.x / G;
  const int co = c0 + cg * SEG;
  float sc[SEG], sh[SEG], cb[SEG], cd[SEG];
#pragma unroll
  for (int i = 0; i < SEG; ++i) {
    const int c = cg * SEG + i;
    sc[i] = tab[c]; sh[i] = tab[CW + c]; cb[i] = tab[2 * CW + c]; cd[i] = tab[3 * CW + c];
  }
  const PixDec pd(y);
  const bool dense = pix_dense(y) && pix_dense(gout) && pix_dense(dy) && (!gres.ptr || pix_dense(gres));
  const int64_t stride = (int64_t)gridDim.x * PPB;
  for (int64_t pix = (int64_t)blockIdx.x * PPB + pl; pix < M; pix += 2 * stride) {      // two pixels in flight per thread
    const int64_t pix2 = pix + stride;
    const bool has2 = pix2 < M;
    const int64_t q = has2 ? pix2 : pix;
    const uint4 ry = ldg16(pptr<T>(y, dense, pd, pix) + co), rg = ldg16(pptr<T>(gout, dense, pd, pix) + co);
    const uint4 ry2 = ldg16(pptr<T>(y, dense, pd, q) + co), rg2 = ldg16(pptr<T>(gout, dense, pd, q) + co);
    uint4 ra = uint4{0u, 0u, 0u, 0u}, ra2 = ra;
    if (gres.ptr && gres_acc) { ra = ldg16(pptr<T>(gres, dense, pd, pix) + co); ra2 = ldg16(pptr<T>(gres, dense, pd, q) + co); }
    float fy[SEG], fg[SEG], fy2[SEG], fg2[SEG], o[SEG], o2[SEG];
    Vec<T>::unpack(ry, fy); Vec<T>::unpack(rg, fg); Vec<T>::unpack(ry2, fy2); Vec<T>::unpack(rg2, fg2);
#pragma unroll
    for (int i = 0; i < SEG; ++i) {
      const float dz = fg[i] * act_grad_f(fmaf(fy[i], sc[i], sh[i]), act);
      const float dz2 = fg2[i] * act_grad_f(fmaf(fy2[i], sc[i], sh[i]), act);
      o[i] = fmaf(sc[i], dz, fmaf(cb[i], fy[i], cd[i]));
      o2[i] = fmaf(sc[i], dz2, fmaf(cb[i], fy2[i], cd[i]));
    }
    stg16(pptr<T>(dy, dense, pd, pix) + co, Vec<T>::pack(o));
    if (has2) stg16(pptr<T>(dy, dense, pd, pix2) + co, Vec<T>::pack(o2));
    if (gres.ptr) {
      if (gres_acc) {
        float a[SEG], a2[SEG];
        Vec<T>::unpack(ra, a); Vec<T>::unpack(ra2, a2);
#pragma unroll
        for (int i = 0; i < SEG; ++i) { fg[i] += a[i]; fg2[i] += a2[i]; }
      }
      stg16(pptr<T>(gres, dense, pd, pix) + co, Vec<T>::pack(fg));
      if (has2) stg16(pptr<T>(gres, dense, pd, pix2) + co, Vec<T>::pack(fg2));
    }
  }
}

// ---- round 6: the whole BatchNorm backward in ONE launch ----
// bn_act_bwd_reduce + bn_act_bwd_apply of a tensor the resident grid holds in registers: every thread loads its NP pixels x 8 channels of
// gout and y ONCE (2*NP 16-byte loads in flight), forms the two per-channel sums from them (shuffles over the pixel lanes of a wave ->
// 2 KB of LDS -> one RETURNING atomic per channel, sum and workgroup into the MYOLO_STAT_COPIES copies of dsum), crosses the grid barrier
// (myolo_dev.h), reads the totals back with sc1 loads and writes dx = sc*dz + cb*y + cd from the SAME registers: gout and y are read once
// instead of twice and one kernel boundary goes away.  Same thread layout, formulas and rounding points as the two kernels above (dz is
// recomputed from the held fp16 values, not kept in fp32), so the results agree with the two-launch form to summation order.
// Grid = ceil(M / (PPB*NP)) x slices workgroups, ALL co-resident (host: <= 512 = two per CU at <= 128 VGPRs and 2 KB of LDS).
// ACT: the activation as a compile-time constant (-1: the run-time switch -- a scalar branch per ELEMENT that also keeps every element's
// intermediate alive: 240 VGPRs at NP = 8 against 128 for the straight-line SiLU form)
template <typename T, int NP, int ACT>
__global__ __launch_bounds__(256) void bn_act_bwd_fused_kernel(myolo_tensor gout, myolo_tensor y, const float* __restrict__ saved,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta, int act_rt,
                                                               float* dsum, float* dgamma, float* dbeta, myolo_tensor dy, myolo_tensor gres,
                                                               int gres_acc, int G, int PPB, BnSplit sp, unsigned int* bar) {
  constexpr int SEG = ET<T>::SEG;
  const int act = ACT >= 0 ? ACT : act_rt;
  __shared__ float red[4 * 2 * 64];      // [wave][2][CW] partial sums, then [4][CW] sc, sh, cb, cd
  const int C = y.c, CW = G * SEG, c0 = blockIdx.y * CW;
  const int cg = threadIdx.x % G, pl = threadIdx.x / G;
  const int co = c0 + cg * SEG;
  const int64_t M = (int64_t)y.n * y.h * y.w;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // (host: every tensor is a channel slice of a dense NHWC buffer -- pixel p sits at ptr + p*sw: one base pointer per tensor and a constant
  //  step per held pixel; the general pptr() kept a 64-bit address per pixel and tensor alive across the barrier: 240 VGPRs at NP = 8)
  // ---- phase 1: the thread's pixels -> registers
  uint4 ry[NP], rg[NP];
  const int64_t p0 = (int64_t)blockIdx.x * NP * PPB + pl;
  const T* yb = reinterpret_cast<const T*>(y.ptr) + p0 * y.sw + co;
  const T* gb = reinterpret_cast<const T*>(gout.ptr) + p0 * gout.sw + co;
  const int64_t ystep = (int64_t)PPB * y.sw, gstep = (int64_t)PPB * gout.sw;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const bool in = p0 + (int64_t)k * PPB < M;               // clamped to the thread's first pixel... of the tensor: the loads stay unconditional
    ry[k] = ldg16(in ? (const void*)(yb + k * ystep) : (const void*)(reinterpret_cast<const T*>(y.ptr) + co));
    rg[k] = ldg16(in ? (const void*)(gb + k * gstep) : (const void*)(reinterpret_cast<const T*>(gout.ptr) + co));
  }
  float s0[SEG], s1[SEG];
  {
    float sc[SEG], sh[SEG], mean[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) {
      const int c = co + i;
      mean[i] = saved[c];
      const float istd = saved[C + c];
      const bool lo = c < sp.cs;
      const int cc = lo ? c : c - sp.cs;
      sc[i] = (lo ? gamma : sp.gamma2)[cc] * istd; sh[i] = (lo ? beta : sp.beta2)[cc] - mean[i] * sc[i];
      s0[i] = 0.f; s1[i] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const float wk = (p0 + (int64_t)k * PPB) < M ? 1.f : 0.f;
      // pixel by pixel: this pixel's packed registers become "new" values that depend on the previous pixel's last accumulate, so hipcc
      // cannot convert all NP pixels to fp32 up front (it did: 255 VGPRs at NP = 8; a sched_barrier alone does not stop it)
      asm volatile("" : "+v"(ry[k].x), "+v"(ry[k].y), "+v"(ry[k].z), "+v"(ry[k].w), "+v"(s0[0]), "+v"(s1[SEG - 1]));
      asm volatile("" : "+v"(rg[k].x), "+v"(rg[k].y), "+v"(rg[k].z), "+v"(rg[k].w), "+v"(s0[0]));
      float fy[SEG], fg[SEG];
      Vec<T>::unpack(ry[k], fy); Vec<T>::unpack(rg[k], fg);
#pragma unroll
      for (int i = 0; i < SEG; ++i) {
        const float dz = wk * fg[i] * act_grad_f(fy[i] * sc[i] + sh[i], act);
        s0[i] += dz;
        s1[i] += dz * (fy[i] - mean[i]);                     // (x invstd below: one multiply per channel instead of one per element)
      }
      __builtin_amdgcn_sched_barrier(0);                     // pixel by pixel: hipcc otherwise unpacks all NP pixels up front (240 VGPRs at NP = 8)
    }
  }
  // pixel lanes of a wave share the channel group: lanes cg, cg + G, ... (G is a power of two <= 32)
  for (int o = G; o < 64; o <<= 1) {
#pragma unroll
    for (int i = 0; i < SEG; ++i) { s0[i] += __shfl_xor(s0[i], o, 64); s1[i] += __shfl_xor(s1[i], o, 64); }
  }
  if (lane < G) {
#pragma unroll
    for (int i = 0; i < SEG; ++i) { red[(wave * 2) * CW + lane * SEG + i] = s0[i]; red[(wave * 2 + 1) * CW + lane * SEG + i] = s1[i]; }
  }
  __syncthreads();
  const int bid = blockIdx.y * gridDim.x + blockIdx.x, nblocks = gridDim.x * gridDim.y;
  float keep = 0.f;
  for (int j = threadIdx.x; j < 2 * CW; j += blockDim.x) {
    const int which = j / CW, cl = j - which * CW;
    float a = (red[(0 * 2 + which) * CW + cl] + red[(1 * 2 + which) * CW + cl]) + (red[(2 * 2 + which) * CW + cl] + red[(3 * 2 + which) * CW + cl]);
    const int c = c0 + cl;
    if (which) a *= saved[C + c];
    // a RETURNING atomic: when its value is back the add has been performed (what the barrier's arrival must not overtake)
    keep += atomicAdd(dsum + (blockIdx.x % MYOLO_STAT_COPIES) * 2 * C + which * C + c, a);
  }
  asm volatile("" ::"v"(keep));
  grid_barrier_xcd(bar, bid, nblocks);
  // the held pixels stay PACKED across the barrier (opaque to the optimiser: reusing phase 1's unpacked fp32 values would double the
  // registers -- 223 instead of ~128 at NP = 8 -- and halve what the resident grid can hold)
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    asm volatile("" : "+v"(ry[k].x), "+v"(ry[k].y), "+v"(ry[k].z), "+v"(ry[k].w));
    asm volatile("" : "+v"(rg[k].x), "+v"(rg[k].y), "+v"(rg[k].z), "+v"(rg[k].w));
  }
  // ---- phase 2: totals -> per-channel constants (as bn_act_bwd_apply), then dx from the registers
  float* tab = red;                      // [4][CW]
  for (int j = threadIdx.x; j < 2 * CW; j += blockDim.x) {       // thread j: statistic j / CW of channel j % CW, summed over the copies
    const int which = j / CW, cl = j - which * CW;
    // (inline-asm rules learnt here: the destination is early-clobber -- the load lands LATER, its register must not double as an address --
    //  and `s_nop 4` precedes the load: hipcc keeps spilled scalar bases in VGPR lanes and restores them with v_readlane right in front of
    //  the asm; a VALU write of an SGPR needs 5 wait states before a VMEM instruction reads it, and the hazard pass does not look inside
    //  inline asm -- without the nops some copies were fetched from the PREVIOUS copy's base: totals off by a few tiles' worth)
    // all copies in flight at once, ONE vector offset + a scalar base per copy (`global_load_dword v, v_off, s[base] sc1`): as plain C++
    // hipcc kept a 64-bit vector address per copy alive (255 VGPRs at NP = 8); inline asm loads are invisible to its waitcnt pass, so the
    // wait is explicit and the results are tied behind it
    const unsigned voff = (unsigned)(which * C + c0 + cl) * 4u;
    float v[MYOLO_STAT_COPIES];
#pragma unroll
    for (int k = 0; k < MYOLO_STAT_COPIES; ++k) {
      const float* base = dsum + (size_t)k * 2 * C;          // uniform
      asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 sc1" : "=&v"(v[k]) : "v"(voff), "s"(base) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < MYOLO_STAT_COPIES; ++k) asm volatile("" : "+v"(v[k]));
    float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < MYOLO_STAT_COPIES; ++k) d[k & 3] += v[k];
    red[4 * CW + j] = (d[0] + d[1]) + (d[2] + d[3]);
  }
  __syncthreads();
  for (int cl = threadIdx.x; cl < CW; cl += blockDim.x) {
    const int c = c0 + cl;
    const float d0 = red[4 * CW + cl], d1 = red[5 * CW + cl];
    const float mean = saved[c], istd = saved[C + c];
    const bool lo = c < sp.cs;
    const int cc = lo ? c : c - sp.cs;
    const float sc = (lo ? gamma : sp.gamma2)[cc] * istd;
    const float Mg = (float)M * (float)sp.world;
    const float k0 = d0 / Mg, k1 = d1 / Mg;
    const float cb = -sc * k1 * istd;
    tab[cl] = sc; tab[CW + cl] = (lo ? beta : sp.beta2)[cc] - mean * sc; tab[2 * CW + cl] = cb; tab[3 * CW + cl] = -sc * k0 - cb * mean;
    if (blockIdx.x == 0) {
      float* dgp = lo ? dgamma : sp.dgamma2; float* dbp = lo ? dbeta : sp.dbeta2;
      const float rw = 1.f / (float)sp.world;
      if (dgp) dgp[cc] += sp.world > 1 ? d1 * rw : d1;
      if (dbp) dbp[cc] += sp.world > 1 ? d0 * rw : d0;
    }
  }
  __syncthreads();
  float sc[SEG], sh[SEG], cb[SEG], cd[SEG];
#pragma unroll
  for (int i = 0; i < SEG; ++i) {
    const int c = cg * SEG + i;
    sc[i] = tab[c]; sh[i] = tab[CW + c]; cb[i] = tab[2 * CW + c]; cd[i] = tab[3 * CW + c];
  }
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int64_t pk = p0 + (int64_t)k * PPB;
    if (pk >= M) continue;
    // (ordered behind the previous pixel's store by the memory clobber: one pixel's fp32 values live at a time)
    asm volatile("" : "+v"(ry[k].x), "+v"(ry[k].y), "+v"(ry[k].z), "+v"(ry[k].w) : : "memory");
    asm volatile("" : "+v"(rg[k].x), "+v"(rg[k].y), "+v"(rg[k].z), "+v"(rg[k].w) : : "memory");
    float fy[SEG], fg[SEG], o[SEG];
    Vec<T>::unpack(ry[k], fy); Vec<T>::unpack(rg[k], fg);
#pragma unroll
    for (int i = 0; i < SEG; ++i) {
      const float dz = fg[i] * act_grad_f(fmaf(fy[i], sc[i], sh[i]), act);
      o[i] = fmaf(sc[i], dz, fmaf(cb[i], fy[i], cd[i]));
    }
    stg16(reinterpret_cast<T*>(dy.ptr) + pk * dy.sw + co, Vec<T>::pack(o));
    if (gres.ptr) {
      T* rp = reinterpret_cast<T*>(gres.ptr) + pk * gres.sw + co;
      if (gres_acc) {
        float a[SEG];
        Vec<T>::unpack(ldg16(rp), a);
#pragma unroll
        for (int i = 0; i < SEG; ++i) fg[i] += a[i];
      }
      stg16(rp, Vec<T>::pack(fg));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

inline bool same_shape(const myolo_tensor* a, const myolo_tensor* b) {
  return a->n == b->n && a->h == b->h && a->w == b->w && a->c == b->c && a->dtype == b->dtype;
}
inline bool vec_ok(const myolo_tensor* t) {
  const int seg = t->dtype == MYOLO_F16 ? 8 : 4;
  return t->ptr && (t->dtype == MYOLO_F16 || t->dtype == MYOLO_F32) && t->c % seg == 0 && t->sw % seg == 0 &&
         t->sh % seg == 0 && t->sn % seg == 0 && ((uintptr_t)t->ptr & 15) == 0;
}

// host view of myolo_bn_split -> kernel argument (no split: cs = C, every channel takes the first parameter set)
inline bool split_ok(const myolo_bn_split* sp, int C, int seg, bool need_params) {
  if (!sp) return true;
  if (sp->count_scale < 0 || sp->c_split <= 0 || sp->c_split > C || sp->c_split % seg) return false;
  if (sp->c_split == C) return true;                         // no second parameter set (count_scale only)
  return !need_params || (sp->gamma2 && sp->beta2);
}
inline BnSplit mk_split(const myolo_bn_split* sp, int C) {
  BnSplit b{};
  b.cs = C;
  b.world = 1;
  if (sp) {
    b.cs = sp->c_split;
    b.world = sp->count_scale > 1 ? sp->count_scale : 1;
    b.gamma2 = sp->gamma2; b.beta2 = sp->beta2; b.rm2 = sp->running_mean2; b.rv2 = sp->running_var2; b.nbt2 = sp->nbt2;
    b.dgamma2 = sp->dgamma2; b.dbeta2 = sp->dbeta2;
  }
  return b;
}


// channel slice per workgroup (see the thread-layout note above): G vector groups x PPB pixels, ny slices
struct BnGeom { int G, PPB, ny; };
inline bool bn_geom(int C, int seg, BnGeom* g) {
  constexpr int slice = 64;          // (round 3 sweep; unsliced: 12.4 us for an 8 MB tensor against 8.5)
  int cw = C;
  if (slice > 0 && slice % seg == 0 && C >= 2 * slice && C % slice == 0) cw = slice;
  g->G = cw / seg;
  if (g->G < 1 || g->G > 256) return false;
  g->PPB = 256 / g->G;
  g->ny = C / cw;
  return true;
}

}  // namespace

extern "C" int myolo_bn_act_fwd_split(const myolo_tensor* y, const float* stats, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, int64_t* nbt, float* saved, float eps,
                                float momentum, int act, const myolo_tensor* res, const myolo_tensor* out,
                                const myolo_bn_split* split, void* stream) {
  if (!y || !out || !vec_ok(y) || !vec_ok(out) || !same_shape(y, out)) return MYOLO_EINVAL;
  if (gamma && (!stats || !beta)) return MYOLO_EINVAL;
  if (split && (!gamma || !split_ok(split, y->c, y->dtype == MYOLO_F16 ? 8 : 4, true))) return MYOLO_EINVAL;
  const BnSplit bs = mk_split(split, y->c);
  myolo_tensor r{};
  if (res && res->ptr) { if (!vec_ok(res) || !same_shape(res, y)) return MYOLO_EINVAL; r = *res; }
  const int seg = y->dtype == MYOLO_F16 ? 8 : 4;
  BnGeom gm;
  if (!bn_geom(y->c, seg, &gm)) return MYOLO_EINVAL;
  const int G = gm.G, PPB = gm.PPB;
  const int64_t M = (int64_t)y->n * y->h * y->w;
  // two pixels per thread and pass; <= 4 workgroups per CU in total (1 for wide UNSLICED layers: each workgroup's prologue sums the
  // MYOLO_STAT_COPIES partial statistics of every channel it covers)
  constexpr int wgs = 1024;
  const int cap = (gm.ny == 1 && y->c >= 512) ? 256 : wgs / gm.ny;
  const dim3 grid(grid_for(M, PPB * 2, cap > 1 ? cap : 1), gm.ny);
  const size_t smem = (size_t)2 * G * seg * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (y->dtype == MYOLO_F16)
    hipLaunchKernelGGL(bn_act_fwd_kernel<half_t>, grid, dim3(G * PPB), smem, st, *y, stats, gamma, beta, running_mean,
                       running_var, nbt, saved, eps, momentum, act, r, *out, G, PPB, bs);
  else
    hipLaunchKernelGGL(bn_act_fwd_kernel<float>, grid, dim3(G * PPB), smem, st, *y, stats, gamma, beta, running_mean,
                       running_var, nbt, saved, eps, momentum, act, r, *out, G, PPB, bs);
  MYOLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int myolo_bn_act_fwd(const myolo_tensor* y, const float* stats, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, int64_t* nbt, float* saved, float eps,
                                float momentum, int act, const myolo_tensor* res, const myolo_tensor* out,
                                void* stream) {
  return myolo_bn_act_fwd_split(y, stats, gamma, beta, running_mean, running_var, nbt, saved, eps, momentum, act, res, out,
                                nullptr, stream);
}

extern "C" int myolo_bn_act_bwd_reduce_split(const myolo_tensor* gout, const myolo_tensor* y, const float* saved,
                                             const float* gamma, const float* beta, int act, float* dsum,
                                             const myolo_bn_split* split, void* stream) {
  if (!gout || !y || !vec_ok(gout) || !vec_ok(y) || !same_shape(gout, y) || !saved || !gamma || !beta || !dsum)
    return MYOLO_EINVAL;
  if (!split_ok(split, y->c, y->dtype == MYOLO_F16 ? 8 : 4, true)) return MYOLO_EINVAL;
  const BnSplit bs = mk_split(split, y->c);
  const int seg = y->dtype == MYOLO_F16 ? 8 : 4;
  BnGeom gm;
  if (!bn_geom(y->c, seg, &gm)) return MYOLO_EINVAL;
  const int G = gm.G, PPB = gm.PPB;
  const int64_t M = (int64_t)y->n * y->h * y->w;
  int gx = (int)((M + PPB * 4 - 1) / (PPB * 4));   // >= 4 pixels per thread (one pass of the 4-deep load pipeline)
  // few, long-lived workgroups: the final per-channel atomics (2 per channel of the slice and workgroup) are same-address
  constexpr int wgs = 512;
  int cap = ((gm.ny == 1 && y->c >= 512) ? 256 : wgs) / gm.ny;
  if (cap < 1) cap = 1;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  const dim3 grid(gx, gm.ny);
  const size_t smem = (size_t)PPB * G * seg * 2 * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (y->dtype == MYOLO_F16)
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<half_t>, grid, dim3(G * PPB), smem, st, *gout, *y, saved, gamma,
                       beta, act, dsum, G, PPB, bs);
  else
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<float>, grid, dim3(G * PPB), smem, st, *gout, *y, saved, gamma,
                       beta, act, dsum, G, PPB, bs);
  MYOLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int myolo_bn_act_bwd_reduce(const myolo_tensor* gout, const myolo_tensor* y, const float* saved,
                                       const float* gamma, const float* beta, int act, float* dsum, void* stream) {
  return myolo_bn_act_bwd_reduce_split(gout, y, saved, gamma, beta, act, dsum, nullptr, stream);
}

extern "C" int myolo_bn_act_bwd_apply_split(const myolo_tensor* gout, const myolo_tensor* y, const float* saved,
                                            const float* gamma, const float* beta, int act, const float* dsum,
                                            float* dgamma, float* dbeta, const myolo_tensor* dy, const myolo_tensor* gres,
                                            int gres_accumulate, const myolo_bn_split* split, void* stream) {
  if (!gout || !y || !dy || !vec_ok(gout) || !vec_ok(y) || !vec_ok(dy) || !same_shape(gout, y) || !same_shape(dy, y))
    return MYOLO_EINVAL;
  if (gamma && (!saved || !beta || !dsum)) return MYOLO_EINVAL;
  if (split && (!gamma || !split_ok(split, y->c, y->dtype == MYOLO_F16 ? 8 : 4, true))) return MYOLO_EINVAL;
  const BnSplit bs = mk_split(split, y->c);
  myolo_tensor r{};
  if (gres && gres->ptr) { if (!vec_ok(gres) || !same_shape(gres, y)) return MYOLO_EINVAL; r = *gres; }
  const int seg = y->dtype == MYOLO_F16 ? 8 : 4;
  BnGeom gm;
  if (!bn_geom(y->c, seg, &gm)) return MYOLO_EINVAL;
  const int G = gm.G, PPB = gm.PPB;
  const int64_t M = (int64_t)y->n * y->h * y->w;
  constexpr int wgs = 1024;
  const int cap = (gm.ny == 1 && y->c >= 512) ? 256 : wgs / gm.ny;
  const dim3 grid(grid_for(M, PPB * 2, cap > 1 ? cap : 1), gm.ny);
  const size_t smem = (size_t)4 * G * seg * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (y->dtype == MYOLO_F16)
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel<half_t>, grid, dim3(G * PPB), smem, st, *gout, *y, saved, gamma, beta,
                       act, dsum, dgamma, dbeta, *dy, r, gres_accumulate, G, PPB, bs);
  else
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel<float>, grid, dim3(G * PPB), smem, st, *gout, *y, saved, gamma, beta,
                       act, dsum, dgamma, dbeta, *dy, r, gres_accumulate, G, PPB, bs);
  MYOLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int myolo_bn_act_bwd_apply(const myolo_tensor* gout, const myolo_tensor* y, const float* saved,
                                      const float* gamma, const float* beta, int act, const float* dsum,
                                      float* dgamma, float* dbeta, const myolo_tensor* dy, const myolo_tensor* gres,
                                      int gres_accumulate, void* stream) {
  return myolo_bn_act_bwd_apply_split(gout, y, saved, gamma, beta, act, dsum, dgamma, dbeta, dy, gres, gres_accumulate, nullptr,
                                      stream);
}

// ---- one-launch BatchNorm backward (round 6) ----
namespace {
static int g_bn_fused = -1;              // MYOLO_BN_BWD_FUSED=0: always the two-launch form
static int g_bn_fused_cap = -1;          // largest grid that is fused (all NP)
inline bool bn_fused_on() {
  if (g_bn_fused < 0) g_bn_fused = getenv("MYOLO_BN_BWD_FUSED") ? atoi(getenv("MYOLO_BN_BWD_FUSED")) : 1;
  return g_bn_fused != 0;
}
// pixels a thread holds: the largest NP whose grid still has >= 256 workgroups (fewer workgroups = a cheaper barrier), bounded by what is
// co-resident at the kernel's register count (173 / 109 / 79 VGPRs at NP = 8 / 4 / 2: two / four / six 256-thread workgroups per CU) AND by
// `bn_fused_cap` (default 256 = one workgroup per CU): the backward shares the chip with the weight-gradient stream, whose workgroups hold
// up to 368 of a SIMD's 512 VGPRs on ~half the CUs, so a 512-workgroup grid waits for them before its barrier can complete -- in the step
// (profiles/r6_bn_fused_step_ab.txt) uncapped 7.76 ms, cap 256 7.70-7.73, cap 128 7.68-7.70 against 7.70-7.73 for the two launches, although
// standalone every fused size wins 1.4-2.9 us (profiles/r6_bn_fused_ubench.txt).  0 = the tensor does not fit the resident grid
inline int bn_fused_np(int64_t M, const BnGeom& gm) {
  if (!bn_fused_on() || (gm.G & (gm.G - 1)) || gm.G > 32) return 0;      // (callers: the slice is at most 64 channels wide)
  static const int caps[3][2] = {{8, 512}, {4, 1024}, {2, 1536}};
  if (g_bn_fused_cap < 0) g_bn_fused_cap = 256;          // (myolo_set_option("bn_fused_cap", n): tests and sweeps)
  const int cap_all = g_bn_fused_cap;
  int best = 0;
  for (auto& c : caps) {
    const int64_t wgs = (M + (int64_t)gm.PPB * c[0] - 1) / ((int64_t)gm.PPB * c[0]) * gm.ny;
    if (wgs > c[1] || wgs > cap_all) continue;
    if (!best) best = c[0];
    if (wgs >= 256) return c[0];
  }
  return best ? 2 : 0;                   // (small tensors: as many workgroups as they give)
}
}  // namespace

int myolo_bn_set(const char* name, int value) {
  if (!strcmp(name, "bn_fused")) { g_bn_fused = value; return 0; }
  if (!strcmp(name, "bn_fused_cap")) { g_bn_fused_cap = value; return 0; }
  return MYOLO_EINVAL;
}

extern "C" int myolo_bn_act_bwd_fused_ok(int dtype, int64_t n_pixels, int c) {
  if (dtype != MYOLO_F16 && dtype != MYOLO_F32) return 0;
  const int seg = dtype == MYOLO_F16 ? 8 : 4;
  BnGeom gm;
  if (c % seg || !bn_geom(c, seg, &gm) || gm.G * seg > 64) return 0;
  return bn_fused_np(n_pixels, gm) > 0;
}

extern "C" int myolo_bn_act_bwd_fused(const myolo_tensor* gout, const myolo_tensor* y, const float* saved, const float* gamma,
                                      const float* beta, int act, float* dsum, float* dgamma, float* dbeta, const myolo_tensor* dy,
                                      const myolo_tensor* gres, int gres_accumulate, const myolo_bn_split* split, uint32_t* barrier,
                                      void* stream) {
  if (!gout || !y || !dy || !vec_ok(gout) || !vec_ok(y) || !vec_ok(dy) || !same_shape(gout, y) || !same_shape(dy, y)) return MYOLO_EINVAL;
  if (!saved || !gamma || !beta || !dsum || !barrier || ((uintptr_t)barrier & 127)) return MYOLO_EINVAL;
  const int seg = y->dtype == MYOLO_F16 ? 8 : 4;
  if (!split_ok(split, y->c, seg, true)) return MYOLO_EINVAL;
  const BnSplit bs = mk_split(split, y->c);
  myolo_tensor r{};
  if (gres && gres->ptr) { if (!vec_ok(gres) || !same_shape(gres, y)) return MYOLO_EINVAL; r = *gres; }
  BnGeom gm;
  const int64_t M = (int64_t)y->n * y->h * y->w;
  auto dense = [](const myolo_tensor& t) { return t.sh == (int64_t)t.w * t.sw && t.sn == (int64_t)t.h * t.sh; };
  const bool all_dense = dense(*gout) && dense(*y) && dense(*dy) && (!r.ptr || dense(r));
  const int np = (all_dense && bn_geom(y->c, seg, &gm) && gm.G * seg <= 64) ? bn_fused_np(M, gm) : 0;
  if (!np) {                             // the tensor does not fit the resident grid: the two-launch form
    int e = myolo_bn_act_bwd_reduce_split(gout, y, saved, gamma, beta, act, dsum, split, stream);
    if (e) return e;
    return myolo_bn_act_bwd_apply_split(gout, y, saved, gamma, beta, act, dsum, dgamma, dbeta, dy, gres, gres_accumulate, split, stream);
  }
  const int G = gm.G, PPB = gm.PPB;
  const dim3 grid((unsigned)((M + (int64_t)PPB * np - 1) / ((int64_t)PPB * np)), gm.ny);
  hipStream_t st = (hipStream_t)stream;
#define BN_FUSED_GO(T_, NP_)                                                                                                          \
  do {                                                                                                                                \
    if (act == MYOLO_ACT_SILU)                                                                                                        \
      hipLaunchKernelGGL((bn_act_bwd_fused_kernel<T_, NP_, MYOLO_ACT_SILU>), grid, dim3(G * PPB), 0, st, *gout, *y, saved, gamma, beta, act, \
                         dsum, dgamma, dbeta, *dy, r, gres_accumulate, G, PPB, bs, barrier);                                            \
    else                                                                                                                              \
      hipLaunchKernelGGL((bn_act_bwd_fused_kernel<T_, NP_, -1>), grid, dim3(G * PPB), 0, st, *gout, *y, saved, gamma, beta, act, dsum,  \
                         dgamma, dbeta, *dy, r, gres_accumulate, G, PPB, bs, barrier);                                                  \
  } while (0)
  if (y->dtype == MYOLO_F16) { if (np == 2) BN_FUSED_GO(half_t, 2); else if (np == 4) BN_FUSED_GO(half_t, 4); else BN_FUSED_GO(half_t, 8); }
  else { if (np == 2) BN_FUSED_GO(float, 2); else if (np == 4) BN_FUSED_GO(float, 4); else BN_FUSED_GO(float, 8); }
#undef BN_FUSED_GO
  MYOLO_CHECK_LAUNCH();
  return 0;
}

// fallback of myolo_conv_desc.bnb: one reduce launch per segment over the gradient the conv launch(es) just stored
int myolo_bnb_fallback(const myolo_conv_desc* d, const myolo_tensor* gout_full, void* stream) {
  if (!d->bnb) return 0;
  const int es = gout_full->dtype == MYOLO_F16 ? 2 : 4;
  for (int i = 0; i < d->nbnb; ++i) {
    const myolo_bn_bwd_seg& s = d->bnb[i];
    myolo_tensor g = *gout_full;
    g.ptr = (char*)gout_full->ptr + (int64_t)s.c0 * es;
    g.c = s.c1 - s.c0;
    const int r = myolo_bn_act_bwd_reduce(&g, &s.y, s.saved, s.gamma, s.beta, s.act, s.dsum, stream);
    if (r) return r;
  }
  return 0;
}
