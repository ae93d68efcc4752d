// load-pattern microbenchmark: how fast can a wave-per-tile streaming kernel read NHWC rows with different lane->address maps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// rows of `pitch` bytes; each wave processes tiles of 16 rows; per tile it reads `rowbytes` (=pitch) bytes of each row.
// mode 0: fragment pattern: lane (r=lane&15, q=lane>>4) reads 16 B at row r, offset kb*64 + q*16   (16 rows x 64 B per instruction)
// mode 1: full-line pattern: lane (r=lane>>3, s=lane&7) reads 16 B at row (i*8 + r), offset s*16 (+128*j)   (8 rows x 128 B per instruction)
// mode 2: linear: the tile's 16*pitch bytes read as one contiguous span (requires dense rows)
template <int MODE, int UNROLL>
__global__ __launch_bounds__(256) void rd(const char* __restrict__ x, int64_t ntiles, int pitch, uint4* out) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  uint4 acc = {0, 0, 0, 0};
  const int per_tile = 16 * pitch / 1024;     // 1 KB wave-instructions per tile
  for (int64_t t = gw; t < ntiles; t += nw * UNROLL) {
    uint4 v[UNROLL][8];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t tt = t + (int64_t)u * nw;
      const char* base = x + tt * 16 * (int64_t)pitch;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i < per_tile && tt < ntiles) {
          const char* p;
          if (MODE == 0) p = base + (int64_t)(lane & 15) * pitch + i * 64 + (lane >> 4) * 16;
          else if (MODE == 1) { const int rows_per = 1024 / pitch > 0 ? 1024 / pitch : 1; (void)rows_per;
            // instruction i covers 8 rows x 128 B: row block (i*8*128/pitch ...) -> generic: byte offset within tile = i*1024 + lane*16 mapped to (row, col) with 128-B granules
            const int64_t off = (int64_t)i * 1024 + lane * 16;      // in units where each row contributes 128 B before moving on
            const int granules_per_row = pitch / 128;
            const int64_t g = off / 128; const int within = off % 128;
            const int64_t row = (g / granules_per_row) ; const int gcol = g % granules_per_row;
            p = base + row * pitch + gcol * 128 + within; }
          else p = base + (int64_t)i * 1024 + lane * 16;
          v[u][i] = *reinterpret_cast<const uint4*>(p);
        } else v[u][i] = uint4{0, 0, 0, 0};
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) { acc.x ^= v[u][i].x; acc.y ^= v[u][i].y; acc.z ^= v[u][i].z; acc.w ^= v[u][i].w; }
  }
  if (acc.x == 0x12345678 && acc.y == 0x9abcdef) out[0] = acc;
}

// store patterns: mode 0: C-fragment-like 8-byte stores (lane (col=lane&15 -> pixel, rowgrp=lane>>4 -> 4 channels)), mode 1: 16-byte row-contiguous
template <int MODE>
__global__ __launch_bounds__(256) void wr(char* __restrict__ y, int64_t ntiles, int pitch) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = gw; t < ntiles; t += nw) {
    char* base = y + t * 16 * (int64_t)pitch;
    if (MODE == 0) {
      for (int nf = 0; nf < pitch / 32; ++nf) {      // 16 channels (32 B) per fragment column block
        uint2 v = {(unsigned)t, (unsigned)lane};
        *reinterpret_cast<uint2*>(base + (int64_t)(lane & 15) * pitch + nf * 32 + (lane >> 4) * 8) = v;
      }
    } else {
      for (int i = 0; i < 16 * pitch / 1024; ++i) {
        uint4 v = {(unsigned)t, (unsigned)lane, 0, 0};
        *reinterpret_cast<uint4*>(base + (int64_t)i * 1024 + lane * 16) = v;
      }
    }
  }
}

int main() {
  const size_t bytes = (size_t)2 << 30;   // 2 GB > 256 MB infinity cache
  char* x; uint4* out;
  CK(hipMalloc(&x, bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(x, 1, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int pitch : {128, 256, 512}) {
    const int64_t ntiles = bytes / (16 * (int64_t)pitch);
    for (int blocks : {2048, 4096, 8192}) {
      auto run = [&](const char* name, auto kern) {
        for (int rep = 0; rep < 2; ++rep) {
          CK(hipEventRecord(e0));
          kern();
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("pitch %4d blocks %5d %-22s %8.1f GB/s\n", pitch, blocks, name, bytes / ms / 1e6);
      };
      run("read frag  u1", [&] { hipLaunchKernelGGL((rd<0, 1>), dim3(blocks), dim3(256), 0, 0, x, ntiles, pitch, out); });
      run("read frag  u2", [&] { hipLaunchKernelGGL((rd<0, 2>), dim3(blocks), dim3(256), 0, 0, x, ntiles, pitch, out); });
      run("read line  u1", [&] { hipLaunchKernelGGL((rd<1, 1>), dim3(blocks), dim3(256), 0, 0, x, ntiles, pitch, out); });
      run("read line  u2", [&] { hipLaunchKernelGGL((rd<1, 2>), dim3(blocks), dim3(256), 0, 0, x, ntiles, pitch, out); });
      run("read linear u2", [&] { hipLaunchKernelGGL((rd<2, 2>), dim3(blocks), dim3(256), 0, 0, x, ntiles, pitch, out); });
      if (blocks == 4096) {
        run("write frag8B", [&] { hipLaunchKernelGGL((wr<0>), dim3(blocks), dim3(256), 0, 0, x, ntiles, pitch); });
        run("write 16B rows", [&] { hipLaunchKernelGGL((wr<1>), dim3(blocks), dim3(256), 0, 0, x, ntiles, pitch); });
      }
    }
  }
  return 0;
}
