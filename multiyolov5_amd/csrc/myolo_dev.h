// Device-side helpers shared by the gfx950 kernels of libmyolo (CDNA4 only: wave64, MFMA, 160 KB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/myolo.h"

typedef _Float16 half_t;
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

// Launch trace (test infrastructure, myolo_trace_start / myolo_trace_read in plan_exec.hip): when on, every launch site records its
// __PRETTY_FUNCTION__ -- for the template launchers that is the kernel family WITH its template arguments (tile shape, ring depth,
// epilogue flags), i.e. exactly which variant the dispatcher picked for a descriptor.  One relaxed load when off.
extern int g_myolo_trace;
void myolo_trace_note(const char* site);
#define MYOLO_CHECK_LAUNCH()                                  \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return (int)e__;                   \
    if (g_myolo_trace) myolo_trace_note(__PRETTY_FUNCTION__); \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize once per kernel instantiation and size (the expansion site's static: a template
// launcher gets one per instantiation).  Calling hipFuncSetAttribute on EVERY launch was a driver round trip per convolution: the
// native executor measured 8.2 us of host time per launch (r3).
#include <atomic>
#define MYOLO_ENSURE_DYN_SMEM(kern, smem)                                                                             \
  do {                                                                                                                \
    static std::atomic<int> cur__{64 * 1024};                                                                         \
    if ((smem) > cur__.load(std::memory_order_relaxed)) {                                                             \
      hipError_t e__ = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (smem)); \
      if (e__ != hipSuccess) return (int)e__;                                                                         \
      cur__.store((smem), std::memory_order_relaxed);                                                                 \
    }                                                                                                                 \
  } while (0)

// element traits: a "segment" is one 16-byte vector (8 halves / 4 floats)
template <typename T> struct ET;
template <> struct ET<half_t> { static constexpr int SEG = 8; static constexpr int KC = 32; };
template <> struct ET<float>  { static constexpr int SEG = 4; static constexpr int KC = 16; };

// hardware exp2 / rcp (v_exp_f32, v_rcp_f32: ~1 ulp each): the IEEE division alone is ~10 VALU slots per element in kernels that
// otherwise move 16 bytes per ~30 instructions
__device__ __forceinline__ float sigmoid_f(float z) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f)); }
__device__ __forceinline__ float silu_f(float z) { return z * sigmoid_f(z); }
// d silu(z)/dz = s(z) * (1 + z * (1 - s(z)))
__device__ __forceinline__ float silu_grad_f(float z) {
  float s = sigmoid_f(z);
  return s * (1.0f + z * (1.0f - s));
}
__device__ __forceinline__ float act_f(float z, int act) {
  return act == MYOLO_ACT_SILU ? silu_f(z) : (act == MYOLO_ACT_SIGMOID ? sigmoid_f(z) : z);
}
__device__ __forceinline__ float act_grad_f(float z, int act) {
  if (act == MYOLO_ACT_SILU) return silu_grad_f(z);
  if (act == MYOLO_ACT_SIGMOID) { float s = sigmoid_f(z); return s * (1.0f - s); }
  return 1.0f;
}

// 16-byte vector <-> float[SEG]
template <typename T> struct Vec;
template <> struct Vec<half_t> {
  static constexpr int N = 8;
  __device__ static void unpack(const uint4& v, float* f) {
    const half_t* h = reinterpret_cast<const half_t*>(&v);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)h[i];
  }
  __device__ static uint4 pack(const float* f) {
    uint4 v;
    half_t* h = reinterpret_cast<half_t*>(&v);
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (half_t)f[i];
    return v;
  }
};
template <> struct Vec<float> {
  static constexpr int N = 4;
  __device__ static void unpack(const uint4& v, float* f) {
    const float* h = reinterpret_cast<const float*>(&v);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = h[i];
  }
  __device__ static uint4 pack(const float* f) {
    uint4 v;
    float* h = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = f[i];
    return v;
  }
};

// explicit global address space: pointers that went through a select (e.g. with the zero page) would otherwise decay to
// FLAT loads, which also count on lgkmcnt and defeat counted vmcnt waits
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) u32x4_t g_u32x4_t;
__device__ __forceinline__ uint4 ldg16(const void* p) {
  const u32x4_t v = *(const g_u32x4_t*)(p);
  return uint4{v.x, v.y, v.z, v.w};
}
typedef __attribute__((address_space(1))) u32x2_t g_u32x2_t;
__device__ __forceinline__ u32x2_t ldg8(const void* p) { return *(const g_u32x2_t*)(p); }
__device__ __forceinline__ void stg16(void* p, const uint4& v) {
  *(g_u32x4_t*)(p) = u32x4_t{v.x, v.y, v.z, v.w};
}

// view addressing (strides in elements)
template <typename T>
__device__ __forceinline__ T* vptr(const myolo_tensor& t, int n, int y, int x) {
  return reinterpret_cast<T*>(t.ptr) + (int64_t)n * t.sn + (int64_t)y * t.sh + (int64_t)x * t.sw;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// 256 zero bytes in device memory.  Out-of-bounds / padded fragment loads are redirected here with an ADDRESS select and the
// load itself stays unconditional: a `cond ? load : 0` select makes hipcc branch around every load and wait vmcnt(0) per
// element, which serialises a software-pipelined K loop (cdna_hip_programming.md, section 5 trap 4c).
__device__ __attribute__((aligned(256))) static unsigned int g_zero_page[64];
__device__ __forceinline__ const char* zero_page() { return reinterpret_cast<const char*>(g_zero_page); }

// ---- strips of a dense channels-last [N,H,W,C] tensor: `npix` consecutive pixels of one row = npix*C contiguous elements.
// Staged through LDS so that HBM sees 16-byte row-contiguous accesses although C (19 classes) is not a vector multiple.
// `gbase` must be 16-byte aligned (callers check (W*C*sizeof(T)) % 16 == 0 and a 256-pixel strip pitch).
template <typename T>
__device__ __forceinline__ void strip_load(const T* gbase, T* lds, int nelem) {
  constexpr int V = 16 / (int)sizeof(T);
  const int nvec = nelem / V;
  for (int v = threadIdx.x; v < nvec; v += blockDim.x)
    reinterpret_cast<uint4*>(lds)[v] = reinterpret_cast<const uint4*>(gbase)[v];
  for (int e = nvec * V + threadIdx.x; e < nelem; e += blockDim.x) lds[e] = gbase[e];
}
template <typename T>
__device__ __forceinline__ void strip_store(T* gbase, const T* lds, int nelem) {
  constexpr int V = 16 / (int)sizeof(T);
  const int nvec = nelem / V;
  for (int v = threadIdx.x; v < nvec; v += blockDim.x)
    reinterpret_cast<uint4*>(gbase)[v] = reinterpret_cast<const uint4*>(lds)[v];
  for (int e = nvec * V + threadIdx.x; e < nelem; e += blockDim.x) gbase[e] = lds[e];
}

static inline int grid_for(int64_t work_items, int threads, int max_blocks = 2048) {
  int64_t b = (work_items + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (int)b;
}

// weight panel [BN][ntaps*cin_pad] of one N tile -> LDS (64-byte blocks, 16-byte segment XOR-ed with H[(row>>2)&3], H = {0,2,3,1}; row
// pitch = odd number of 64-byte blocks): U independent 16-byte loads are in flight per thread before the first LDS store (a
// load -> store loop serialises on one L2 round trip per iteration: ~1 us x 18 iterations for a 3x3 64->64 panel)
// PERM: panel row R = nf*16 + j holds output channel panel_chan(R) of the N tile instead of channel R.  With D^T = W . X^T a lane holds
// rows 4*lq .. 4*lq+3 of every 16-row fragment; under this order the rows it holds in fragments 2q and 2q+1 are the 8 CONSECUTIVE
// channels 32q + 8*lq .. + 7 of its pixel (16-byte stores).  BN must be a multiple of 32.
__host__ __device__ __forceinline__ int panel_chan(int R) { return (R >> 5) * 32 + 8 * ((R & 15) >> 2) + 4 * ((R >> 4) & 1) + (R & 3); }
template <int BN, int NT, bool PERM = false>
__device__ __forceinline__ void stage_weight_panel(char* sB, const char* w, int tn, int pitchB, int cin_pad, int ntaps, int wtaps,
                                                   const int* tap_w, int tid) {
  static_assert(!PERM || BN % 32 == 0, "permuted panels pair 16-row fragments");
  constexpr int U = 8;
  const int vec_per_tap = cin_pad / 8;
  const int vec_per_row = ntaps * vec_per_tap;
  const int total = BN * vec_per_row;
  for (int base = 0; base < total; base += NT * U) {
    uint4 tmp[U];
    int dst[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int v = base + u * NT + tid;
      const bool ok = v < total;
      const int vv = ok ? v : 0;
      const int r = vv / vec_per_row; const int q = vv - r * vec_per_row;
      const int t = q / vec_per_tap; const int s2 = q - t * vec_per_tap;
      tmp[u] = ldg16(w + ((int64_t)((tn * BN + (PERM ? panel_chan(r) : r)) * wtaps + tap_w[t]) * cin_pad + s2 * 8) * 2);
      const int g = t * vec_per_tap + s2;
      const int sw = (0x78 >> (((r >> 2) & 3) * 2)) & 3;
      dst[u] = ok ? r * pitchB + (g >> 2) * 64 + (((g & 3) ^ sw) << 4) : -1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (dst[u] >= 0) *reinterpret_cast<uint4*>(sB + dst[u]) = tmp[u];
  }
}

// ---- 8 consecutive fp16 channels of one pixel (the epilogues of the D^T = W . X^T kernels) ----
__device__ __forceinline__ u32x4_t pack_h8(const float* v) {
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  u32x4_t o;
  uint32_t* ow = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h2v hv; hv.x = (_Float16)v[2 * i]; hv.y = (_Float16)v[2 * i + 1];
    ow[i] = *reinterpret_cast<const uint32_t*>(&hv);
  }
  return o;
}
__device__ __forceinline__ void add_h8(float* v, const u32x4_t& g) {
  const _Float16* gh = reinterpret_cast<const _Float16*>(&g);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] += (float)gh[i];
}
// buffer load of a lane's 8 channels: whole (ok8), the first four only (ok4; issued only when the layer has such a group), or zeros
__device__ __forceinline__ u32x4_t buf_load_h8(__amdgpu_buffer_rsrc_t rs, int off, bool ok8, bool ok4, bool half_tail) {
  constexpr int OOB_ = 0x7fff0000;
  u32x4_t g = __builtin_amdgcn_raw_buffer_load_b128(rs, ok8 ? off : OOB_, 0, 0);
  if (half_tail) {
    const u32x2_t g2 = __builtin_amdgcn_raw_buffer_load_b64(rs, ok4 ? off : OOB_, 0, 0);
    g.x |= g2.x; g.y |= g2.y;
  }
  return g;
}

// ---- device-wide barrier inside a launch (round 6; MI355X_MICROARCH.md row "barrier-xcd", measured here: profiles/r6_grid_barrier_ubench.txt:
// 4.7 / 6.5 / 10.5 us at 256 / 512 / 1024 workgroups without a release fence, 6.2 / 9.8 / 17.5 with one per workgroup) ----
// State: MYOLO_GRID_BARRIER_WORDS uint32 zeroed ONCE by the caller (include/myolo.h); every word sits on its own 128-byte line:
//   [g]*32, g = 0..7: arrival counter of group g = linear block id & 7 (the XCD the block is observed to run on -- a LOGICAL group, nothing
//   depends on the placement), [8+g]*32: generation word of group g, [16]*32: group-leader counter, [17]*32: top generation,
//   [18]*32: sticky timeout flag.  The last arriver zeroes the counter before it publishes the generation, so launches on ONE stream
//   may share a state block without a memset; two launches in flight at once need two blocks.
// Visibility contract: what crosses the barrier must be written with agent-scope atomics (RETURNING ones whose result the writer has
// consumed: the RMW has been performed before the arrival) or sc1 stores, and read behind the barrier with __hip_atomic_load(agent)
// (sc1 loads): then no release / acquire fence is needed (cdna_hip_programming.md Guideline 16, form R1).  Residency: EVERY workgroup
// of the grid must be co-resident (callers size the grid for one or two workgroups per CU); every spin is bounded, a timeout sets the flag
// and lets the kernel finish with garbage instead of hanging the device.
#define MYOLO_GRID_BARRIER_WORDS (19 * 32)
typedef __attribute__((address_space(1))) unsigned int gbar_u32;
__device__ __forceinline__ bool gbar_spin(gbar_u32* p, unsigned old, gbar_u32* tmo) {
  for (unsigned spins = 0; spins < (1u << 19); ++spins) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != old) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}
// FENCED: lane 0 also issues an agent-scope release before it arrives and an acquire after the wait (plain stores / plain loads may then cross
// the barrier; +1.5 us at 256 workgroups, profiles/r6_grid_barrier_ubench.txt rows `xcd` vs `xcd_lead`).  The library's two users publish
// through returning atomics and read back with sc1 loads, which needs neither.
template <bool FENCED = false>
__device__ __forceinline__ void grid_barrier_xcd(unsigned int* state, int bid, int nblocks) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every wave: its atomics / sc1 stores have been performed
  __syncthreads();
  if (threadIdx.x == 0) {
    gbar_u32* w = (gbar_u32*)state;
    const int g = bid & 7;
    const int ng = (nblocks - g + 7) >> 3;                 // blocks with id & 7 == g
    const int ngroups = nblocks < 8 ? nblocks : 8;
    const unsigned gen0 = __hip_atomic_load(w + (8 + g) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (FENCED) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");    // (buffer_wbl2 sc1: what this workgroup wrote leaves its XCD's L2)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (ROCm 7.2 may drop the wait behind the write-back: restated where the compiler cannot)
    }
    const unsigned old = __hip_atomic_fetch_add(w + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)ng - 1) {                         // group leader = last arriver of the group
      __hip_atomic_store(w + g * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned tgen0 = __hip_atomic_load(w + 17 * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned t = __hip_atomic_fetch_add(w + 16 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == (unsigned)ngroups - 1) {
        __hip_atomic_store(w + 16 * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(w + 17 * 32, tgen0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        gbar_spin(w + 17 * 32, tgen0, w + 18 * 32);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(w + (8 + g) * 32, gen0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      gbar_spin(w + (8 + g) * 32, gen0, w + 18 * 32);
    }
    if (FENCED) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (buffer_inv sc1: nothing this CU cached before the barrier is served afterwards)
  }
  __syncthreads();
}
__device__ __forceinline__ float ld_agent_f32(const float* p) {
  return __hip_atomic_load((const __attribute__((address_space(1))) float*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// hipMemsetAsync's replacement inside the library (csrc/plan_exec.hip): up to two regions of 32-bit words in ONE kernel launch.  rocclr fill
// commands cost ~5 us each even for 16 bytes and do not pack behind kernels; as NODES of a captured graph they were seen to stop clearing
// (zero_fill_kernel).  Regions: 4-byte aligned, sizes multiples of 4, 16-byte aligned bases beyond 1 KB.
int myolo_fill_words2(void* a, size_t a_bytes, unsigned int av, void* b, size_t b_bytes, unsigned int bv, hipStream_t st);

// ---- BatchNorm-backward statistics in a dgrad epilogue (myolo_conv_desc.bnb) ----
struct BnbSeg {              // device-side copy of one myolo_bn_bwd_seg
  int c0, c1;
  const char* y; int64_t y_sn, y_sh, y_sw;
  const float* saved; const float* gamma; const float* beta; float* dsum; int act;
};
struct BnbArgs { int n; BnbSeg seg[MYOLO_MAX_BNB]; };
// host: true when every segment starts and ends on a multiple of `bn` (so that one N tile belongs to one segment)
static inline bool bnb_aligned(const myolo_conv_desc* d, int bn) {
  for (int i = 0; i < d->nbnb; ++i)
    if (d->bnb[i].c0 % bn || (d->bnb[i].c1 % bn && d->bnb[i].c1 != d->y.c)) return false;
  return true;
}
static inline void bnb_fill(BnbArgs* a, const myolo_conv_desc* d) {
  a->n = d->bnb ? d->nbnb : 0;
  for (int i = 0; i < a->n; ++i) {
    const myolo_bn_bwd_seg& s = d->bnb[i];
    a->seg[i] = BnbSeg{s.c0, s.c1, (const char*)s.y.ptr, s.y.sn, s.y.sh, s.y.sw, s.saved, s.gamma, s.beta, s.dsum, s.act};
  }
}
// bn_act.hip: the reduce pass over the STORED gout for every segment of d (fallback when the conv kernel cannot fold it)
int myolo_bnb_fallback(const myolo_conv_desc* d, const myolo_tensor* gout_full, void* stream);

// bn_act.hip: "bn_fused" (0 off / 1 on), "bn_fused_cap" (largest resident grid of the one-launch BatchNorm backward)
int myolo_bn_set(const char* name, int value);
int myolo_pool_set(const char* name, int value);
int myolo_stem_set(const char* name, int value);      // stem_wgrad.hip: "stem_wgrad" (0 off), "stem_ks" (workgroups)      // pool_resize.hip: "spp_naive"
// conv_stream.hip: streaming variant of myolo_conv; -1 = layer does not qualify
int myolo_conv_stream_try(const myolo_conv_desc* d, void* stream, int* bnb_done);
// conv_halo.hip: LDS-staged input tiles for k x k stride-1 convolutions; -1 = layer does not qualify
int myolo_conv_halo_try(const myolo_conv_desc* d, void* stream, int* bnb_done);
int myolo_conv_halo_set(const char* name, int value);
int myolo_conv_igemm_set(const char* name, int value);   // conv_igemm.hip: "igemm_bm", "igemm_bn", "igemm_wgs" (tile-shape experiments)
// conv_small.hip: split-K kernel for small maps (eval epilogues); -1 = layer does not qualify
int myolo_conv_small_try(const myolo_conv_desc* d, void* stream);
int myolo_conv_small_set(const char* name, int value);
// conv_mid.hip: LDS-DMA staged 128-byte K steps, 8 waves per tile, for the training convolutions of the mid-size / small maps;
// -1 = layer does not qualify.  myolo_conv_mid_mode(): 0 off, 1 instead of conv_igemm, 2 ahead of the halo / streaming kernels too
int myolo_conv_mid_try(const myolo_conv_desc* d, void* stream, int* bnb_done);
int myolo_conv_mid_set(const char* name, int value);
int myolo_conv_mid_mode();
// conv_pair.hip: "pair_mode" (0: always two launches), "pair_th" (tile height 4 / 8 forced; tests)
int myolo_conv_pair_set(const char* name, int value);
// conv_midx.hip: conv_mid with the input of a k x k stride-1 layer resident in LDS as a halo tile; -1 = layer does not qualify
int myolo_conv_midx_try(const myolo_conv_desc* d, void* stream);
int myolo_conv_midx_set(const char* name, int value);
// conv_wgrad_tile.hip: weight gradient over LDS-staged spatial tiles; -1 = layer does not qualify
int myolo_wgrad_tile_try(const myolo_wgrad_desc* d, void* stream, int* out_ks, int* out_cop, int* out_cip, int* used_ws);
int myolo_wgrad_tile_set(const char* name, int value);
