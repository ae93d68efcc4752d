// Grid-barrier microbenchmark (round 6): what does a device-wide barrier INSIDE a launch cost on this part?
// Three designs, each checked for correctness (L1-warm neighbour re-reads, uneven arrival) and timed as
// (t(kernel with R barriers) - t(kernel with 1 barrier)) / (R - 1) at 256 / 512 / 1024 workgroups of 256 threads:
//   flat    : one arrival counter + one generation word; every workgroup release-fences before it arrives, acquires after
//   xcd     : MI355X_MICROARCH.md row "barrier-xcd": 8 arrival counters (group = blockIdx & 7 -- the XCD the block is observed to run
//             on, a LOGICAL group so nothing depends on the placement), the last arriver of a group goes to a top counter, the last
//             group leader bumps the top generation, every leader then bumps its group's generation word; every workgroup
//             release-fences its own stores (placement independent) and acquires after the wait
//   xcd_lead: the same, but only the group leader issues the release fence (buffer_wbl2 of the XCD's L2): correct ONLY while
//             blockIdx & 7 really is the XCD -- reported for the price, not used by the library
// All state is self-resetting (the last arriver zeroes the counter before it publishes the generation), so a launch needs no memset.
// Every spin is bounded; a timeout sets a flag word that the host prints.
// build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef __attribute__((address_space(1))) unsigned int gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct BarState {           // every word on its own 128-byte line
  unsigned int* w;          // [0..7]*32: group counters, [8..15]*32: group generations, 16*32: top counter, 17*32: top generation, 18*32: timeout flag
};
__device__ __forceinline__ gu32* word(const BarState& s, int i) { return (gu32*)(s.w + i * 32); }

__device__ __forceinline__ bool spin_until_changed(gu32* p, unsigned old, gu32* tmo) {
  for (unsigned spins = 0; spins < (1u << 22); ++spins) {
    if (__hip_atomic_load(p, RLX_AGENT) != old) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  __hip_atomic_store(tmo, 1u, RLX_AGENT);
  return false;
}

// MODE 0 flat, 1 xcd, 2 xcd_lead
template <int MODE>
__device__ __forceinline__ void grid_barrier(const BarState& s, int nblocks) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every wave: its stores have left the CU
  __syncthreads();
  if (threadIdx.x == 0) {
    gu32* tmo = word(s, 18);
    if (MODE == 0) {
      const unsigned gen0 = __hip_atomic_load(word(s, 8), RLX_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned old = __hip_atomic_fetch_add(word(s, 0), 1u, RLX_AGENT);
      if (old == (unsigned)nblocks - 1) {
        __hip_atomic_store(word(s, 0), 0u, RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(word(s, 8), gen0 + 1, RLX_AGENT);
      } else {
        spin_until_changed(word(s, 8), gen0, tmo);
      }
    } else {
      const int g = blockIdx.x & 7;
      const int ng = (nblocks - g + 7) >> 3;                 // blocks with blockIdx & 7 == g
      const int ngroups = nblocks < 8 ? nblocks : 8;
      const unsigned gen0 = __hip_atomic_load(word(s, 8 + g), RLX_AGENT);
      if (MODE == 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      const unsigned old = __hip_atomic_fetch_add(word(s, g), 1u, RLX_AGENT);
      if (old == (unsigned)ng - 1) {                         // group leader = last arriver of the group
        __hip_atomic_store(word(s, g), 0u, RLX_AGENT);
        const unsigned tgen0 = __hip_atomic_load(word(s, 17), RLX_AGENT);
        if (MODE == 2) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned t = __hip_atomic_fetch_add(word(s, 16), 1u, RLX_AGENT);
        if (t == (unsigned)ngroups - 1) {
          __hip_atomic_store(word(s, 16), 0u, RLX_AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(word(s, 17), tgen0 + 1, RLX_AGENT);
        } else {
          spin_until_changed(word(s, 17), tgen0, tmo);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(word(s, 8 + g), gen0 + 1, RLX_AGENT);
      } else {
        spin_until_changed(word(s, 8 + g), gen0, tmo);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// R barriers; between barriers every block writes a 128-byte record (phase-tagged) and after the barrier re-reads the record of
// another block (which it also read BEFORE the barrier: L1-warm consumer) and counts stale words.  `skew`: blocks whose index is a
// multiple of 7 burn time before arriving (uneven load).
template <int MODE>
__global__ __launch_bounds__(256) void bar_kernel(BarState s, unsigned* rec, int R, int skew, unsigned* bad) {
  const int n = gridDim.x, b = blockIdx.x, t = threadIdx.x;
  const int peer = (b + 37) % n;
  unsigned nbad = 0;
  for (int r = 0; r < R; ++r) {
    if (t < 32) rec[b * 32 + t] = (unsigned)(r + 1) * 1000003u + b * 32 + t;
    unsigned warm = t < 32 ? rec[peer * 32 + t] : 0;            // pull the peer's (old) line into this CU's L1
    if (skew && (b % 7) == 0) { for (int i = 0; i < skew; ++i) __builtin_amdgcn_s_sleep(64); }
    grid_barrier<MODE>(s, n);
    if (t < 32) {
      const unsigned v = rec[peer * 32 + t];
      if (v != (unsigned)(r + 1) * 1000003u + peer * 32 + t) ++nbad;
      if (warm == 0xdeadbeefu) ++nbad;
    }
    grid_barrier<MODE>(s, n);        // the record may be rewritten only after every reader is done
  }
  if (nbad) atomicAdd(bad, nbad);
}

// timing kernel: R barriers back to back, nothing published
template <int MODE>
__global__ __launch_bounds__(256) void bar_time(BarState s, int R) {
  for (int r = 0; r < R; ++r) grid_barrier<MODE>(s, gridDim.x);
}

template <int MODE>
static double time_bar(BarState s, int nblocks, int R, hipEvent_t e0, hipEvent_t e1) {
  std::vector<float> ts;
  for (int it = 0; it < 12; ++it) {
    CK(hipEventRecord(e0));
    bar_time<MODE><<<nblocks, 256>>>(s, R);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 2) ts.push_back(ms * 1000.f);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

template <int MODE>
static void run(const char* name, BarState s, unsigned* rec, unsigned* bad, hipEvent_t e0, hipEvent_t e1) {
  for (int nblocks : {256, 512, 1024}) {
    CK(hipMemset(bad, 0, 4));
    bar_kernel<MODE><<<nblocks, 256>>>(s, rec, 50, 0, bad);
    bar_kernel<MODE><<<nblocks, 256>>>(s, rec, 50, 20, bad);
    CK(hipDeviceSynchronize());
    unsigned hbad = 0, tmo = 0;
    CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&tmo, s.w + 18 * 32, 4, hipMemcpyDeviceToHost));
    const double t1 = time_bar<MODE>(s, nblocks, 1, e0, e1);
    const double t101 = time_bar<MODE>(s, nblocks, 101, e0, e1);
    const double t21 = time_bar<MODE>(s, nblocks, 21, e0, e1);
    printf("%-9s wgs %4d  per barrier %6.2f us (R=101) %6.2f us (R=21)   kernel with 1 barrier %6.2f us   stale words %u  timeout %u\n",
           name, nblocks, (t101 - t1) / 100.0, (t21 - t1) / 20.0, t1, hbad, tmo);
  }
}

__global__ void empty_kernel() {}

int main() {
  BarState s;
  CK(hipMalloc(&s.w, 32 * 32 * 4));
  CK(hipMemset(s.w, 0, 32 * 32 * 4));
  unsigned *rec, *bad;
  CK(hipMalloc(&rec, 1024 * 32 * 4)); CK(hipMemset(rec, 0, 1024 * 32 * 4));
  CK(hipMalloc(&bad, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, bar_time<1>, 256, 0));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s, %d CUs, occupancy API %d blocks of 256 threads per CU\n", p.gcnArchName, p.multiProcessorCount, occ);
  // launch overhead reference: an empty kernel, event-timed
  { std::vector<float> ts; for (int it = 0; it < 12; ++it) { CK(hipEventRecord(e0)); empty_kernel<<<256, 256>>>(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it >= 2) ts.push_back(ms * 1000.f); }
    std::sort(ts.begin(), ts.end()); printf("empty 256-block kernel between two events: %.2f us\n", ts[ts.size() / 2]); }
  run<0>("flat", s, rec, bad, e0, e1);
  run<1>("xcd", s, rec, bad, e0, e1);
  run<2>("xcd_lead", s, rec, bad, e0, e1);
  return 0;
}
