// Layout transforms at the edge of the hot path (HBM-bound, one pass each):
//   * OIHW master weights -> packed [row][tap][col] MFMA operand (forward or transposed for dgrad),
//     optional per-output-channel scale (BN folding, reference utils/torch_utils.py:193-195);
//   * Focus space-to-depth + image cast (reference models/common.py:550, train.py:342 `/255`).
#include "myolo_dev.h"

namespace {

template <typename S, typename D>
__global__ void pack_weight_kernel(const S* __restrict__ src, D* __restrict__ dst, int cout, int cin, int ntaps,
                                   int rows_pad, int cols_pad, int transpose, const float* __restrict__ row_scale) {
  const int64_t total = (int64_t)rows_pad * ntaps * cols_pad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % cols_pad);
    const int t = (int)((i / cols_pad) % ntaps);
    const int row = (int)(i / ((int64_t)cols_pad * ntaps));
    const int co = transpose ? col : row, ci = transpose ? row : col;
    float v = 0.f;
    if (co < cout && ci < cin) {
      v = (float)src[((int64_t)co * cin + ci) * ntaps + t];
      if (row_scale) v *= row_scale[co];
    }
    dst[i] = (D)v;
  }
}

// every weight of a training plan in ONE launch (the per-conv launches were ~160 x 5 us per step): job table (device int64
// [njobs][12]) = {src, dst, cout, cin, ntaps, rows_pad, cols_pad, transpose, src_dtype, dst_dtype, src2, cout2}; chunk list (device
// int32 [nchunks][2]) = {job, first packed element}.  src2 (or 0): a second OIHW tensor stacked behind src along cout (output
// channels cout .. cout+cout2: two convolutions of one input run as one launch)
__global__ __launch_bounds__(256) void pack_weights_mt_kernel(const int64_t* jobs, const int32_t* chunks, int chunk_elems) {
  const int j = chunks[blockIdx.x * 2], start = chunks[blockIdx.x * 2 + 1];
  const int64_t* e = jobs + (int64_t)j * 12;
  const void* src = reinterpret_cast<const void*>(e[0]);
  void* dst = reinterpret_cast<void*>(e[1]);
  const int cout = (int)e[2], cin = (int)e[3], ntaps = (int)e[4], rows_pad = (int)e[5], cols_pad = (int)e[6];
  const int transpose = (int)e[7], sdt = (int)e[8], ddt = (int)e[9];
  const void* src2 = reinterpret_cast<const void*>(e[10]);
  const int cout2 = src2 ? (int)e[11] : 0;
  // (one packed weight tensor is far below 2^31 elements: 32-bit index arithmetic, one division chain per thread, then strides)
  const uint32_t total = (uint32_t)rows_pad * (uint32_t)ntaps * (uint32_t)cols_pad;
  uint32_t end = (uint32_t)start + (uint32_t)chunk_elems;
  if (end > total) end = total;
  const uint32_t ucols = (uint32_t)cols_pad, utaps = (uint32_t)ntaps;
  const bool f16s = sdt == MYOLO_F16, f16d = ddt == MYOLO_F16;
  for (uint32_t i = (uint32_t)start + threadIdx.x; i < end; i += 256) {
    const uint32_t q = i / ucols;
    const int col = (int)(i - q * ucols);
    const uint32_t row_u = q / utaps;
    const int t = (int)(q - row_u * utaps);
    const int row = (int)row_u;
    const int co = transpose ? col : row, ci = transpose ? row : col;
    float v = 0.f;
    if (co < cout + cout2 && ci < cin) {
      const bool second = co >= cout;
      const uint32_t si = ((uint32_t)(second ? co - cout : co) * (uint32_t)cin + (uint32_t)ci) * utaps + (uint32_t)t;
      const void* sp = second ? src2 : src;
      v = f16s ? (float)((const half_t*)sp)[si] : ((const float*)sp)[si];
    }
    if (f16d) ((half_t*)dst)[i] = (half_t)v; else ((float*)dst)[i] = v;
  }
}

// the same through LDS tiles (chunk_elems == 0: chunks = {job, tile}): the element-per-thread kernel above reads the OIHW source
// with the destination's index order -- a transposed (dgrad) pack touches a different 64-byte line per lane, a 3x3 forward pack
// re-reads every line once per tap: 414 MB of fetches for 31 MB of weights, 84 us at the head of every training forward.  Here a
// workgroup owns a [co tile] x [ci tile] x all taps block: the source rows are read as contiguous runs (ci*taps is contiguous in
// OIHW), transposed in LDS, and written as 64-byte runs of the destination's fastest index.  Padding rows / columns are NOT written:
// the packed tensors are allocated zeroed and nothing else writes them.
constexpr int PACK_TILE = 512;             // (co x ci) pairs per tile: 16 x 32 (forward operand) or 32 x 16 (transposed); 64 x 64 for 1x1 weights
// NT / TR: compile-time tap count and orientation (1x1 and 3x3 weights: the index arithmetic divides by constants); NT = 0: any
// tap count, read from the job.  The first version took the divisors from the job row for every element and always reserved the
// 25-tap tile (51 KB: three workgroups per CU): 61 us for 45 MB at the head of every training forward.
template <int NT, bool TR>
__device__ __forceinline__ void pack_tile_body(float* tile, const int64_t* e, int tid_tile, int ntaps_rt, int transpose_rt) {
  const void* src = reinterpret_cast<const void*>(e[0]);
  void* dst = reinterpret_cast<void*>(e[1]);
  const int cout = (int)e[2], cin = (int)e[3], cols_pad = (int)e[6];
  const int sdt = (int)e[8], ddt = (int)e[9];
  const void* src2 = reinterpret_cast<const void*>(e[10]);
  const int cout_all = cout + (src2 ? (int)e[11] : 0);
  const int ntaps = NT ? NT : ntaps_rt;
  const bool transpose = NT ? TR : transpose_rt != 0;
  const int tco = ntaps == 1 ? 64 : (transpose ? 32 : 16), tci = ntaps == 1 ? 64 : (transpose ? 16 : 32);   // ~4-13 K elements per workgroup
  const int tiles_ci = (cin + tci - 1) / tci;
  const int co0 = (tid_tile / tiles_ci) * tco, ci0 = (tid_tile % tiles_ci) * tci;
  const bool f16s = sdt == MYOLO_F16, f16d = ddt == MYOLO_F16;
  const int run = tci * ntaps;                                   // contiguous source elements per output channel of the tile
  const int nci = cin - ci0 < tci ? cin - ci0 : tci;
  const int live = nci * ntaps;
  for (int i = threadIdx.x; i < tco * run; i += 256) {
    const int col = i / run, r = i - col * run;
    const int co = co0 + col;
    float v = 0.f;
    if (co < cout_all && r < live) {
      const bool second = co >= cout;
      const void* sp = second ? src2 : src;
      const int64_t si = ((int64_t)(second ? co - cout : co) * cin + ci0) * ntaps + r;
      v = f16s ? (float)((const half_t*)sp)[si] : ((const float*)sp)[si];
    }
    tile[i] = v;
  }
  __syncthreads();
  // destination element (row, t, col): forward operand row = co, col = ci; transposed row = ci, col = co
  const int fast = transpose ? tco : tci, slow = transpose ? tci : tco;
  for (int i = threadIdx.x; i < slow * ntaps * fast; i += 256) {
    const int f = i % fast, q = i / fast;
    const int t = q % ntaps, sl = q / ntaps;
    const int col = transpose ? f : sl, cil = transpose ? sl : f;       // tile-local co / ci
    const int co = co0 + col, ci = ci0 + cil;
    if (co >= cout_all || ci >= cin) continue;
    const float v = tile[col * run + cil * ntaps + t];
    const int64_t di = transpose ? ((int64_t)ci * ntaps + t) * cols_pad + co : ((int64_t)co * ntaps + t) * cols_pad + ci;
    if (f16d) ((half_t*)dst)[di] = (half_t)v; else ((float*)dst)[di] = v;
  }
}

__global__ __launch_bounds__(256) void pack_weights_tiled_kernel(const int64_t* __restrict__ jobs, const int32_t* __restrict__ chunks) {
  extern __shared__ float tile[];          // [tco][tci * taps]
  const int j = chunks[blockIdx.x * 2], tid_tile = chunks[blockIdx.x * 2 + 1];
  const int64_t* e = jobs + (int64_t)j * 12;
  const int ntaps = (int)e[4], transpose = (int)e[7];
  if (ntaps == 1) { if (transpose) pack_tile_body<1, true>(tile, e, tid_tile, 1, 1); else pack_tile_body<1, false>(tile, e, tid_tile, 1, 0); }
  else if (ntaps == 9) { if (transpose) pack_tile_body<9, true>(tile, e, tid_tile, 9, 1); else pack_tile_body<9, false>(tile, e, tid_tile, 9, 0); }
  else pack_tile_body<0, false>(tile, e, tid_tile, ntaps, transpose);
}

template <typename S, typename D>
__global__ void focus_pack_kernel(const S* __restrict__ img, int n, int h, int w, float mul, myolo_tensor out) {
  const int ho = h >> 1, wo = w >> 1;
  const int64_t total = (int64_t)n * ho * wo;
  const int64_t plane = (int64_t)h * w;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % wo);
    const int oy = (int)((i / wo) % ho);
    const int b = (int)(i / ((int64_t)wo * ho));
    D* o = vptr<D>(out, b, oy, ox);
    const S* base = img + (int64_t)b * 3 * plane + (int64_t)(2 * oy) * w + 2 * ox;
    D vals[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int dy = q & 1, dx = q >> 1;          // q: (0,0),(1,0),(0,1),(1,1) as (row parity, col parity)
#pragma unroll
      for (int c = 0; c < 3; ++c) vals[3 * q + c] = (D)((float)base[c * plane + (int64_t)dy * w + dx] * mul);
    }
#pragma unroll
    for (int c = 12; c < 16; ++c) vals[c] = (D)0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) o[c] = vals[c];
  }
}

}  // namespace

extern "C" int myolo_version(void) { return 1; }
extern "C" const char* myolo_arch(void) { return "gfx950"; }

extern "C" int myolo_pack_weight(const void* w, int src_dtype, int cout, int cin, int kh, int kw, void* dst,
                                 int dst_dtype, int rows_pad, int cols_pad, int transpose, const float* row_scale,
                                 void* stream) {
  if (!w || !dst || cout <= 0 || cin <= 0) return MYOLO_EINVAL;
  const int ntaps = kh * kw;
  if ((transpose ? cin : cout) > rows_pad || (transpose ? cout : cin) > cols_pad) return MYOLO_EINVAL;
  const int64_t total = (int64_t)rows_pad * ntaps * cols_pad;
  const int grid = grid_for(total, 256);
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH(S, D)                                                                                           \
  hipLaunchKernelGGL((pack_weight_kernel<S, D>), dim3(grid), dim3(256), 0, st, (const S*)w, (D*)dst, cout, cin, \
                     ntaps, rows_pad, cols_pad, transpose, row_scale)
  if (src_dtype == MYOLO_F32 && dst_dtype == MYOLO_F16) LAUNCH(float, half_t);
  else if (src_dtype == MYOLO_F32 && dst_dtype == MYOLO_F32) LAUNCH(float, float);
  else if (src_dtype == MYOLO_F16 && dst_dtype == MYOLO_F16) LAUNCH(half_t, half_t);
  else if (src_dtype == MYOLO_F16 && dst_dtype == MYOLO_F32) LAUNCH(half_t, float);
  else return MYOLO_EINVAL;
#undef LAUNCH
  MYOLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int myolo_focus_pack(const void* img, int src_dtype, int n, int h, int w, float mul,
                                const myolo_tensor* out, void* stream) {
  if (!img || !out || !out->ptr || (h & 1) || (w & 1) || out->c != 16 || out->h != h / 2 || out->w != w / 2 ||
      out->n != n)
    return MYOLO_EINVAL;
  const int grid = grid_for((int64_t)n * (h / 2) * (w / 2), 256, 4096);
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH(S, D) hipLaunchKernelGGL((focus_pack_kernel<S, D>), dim3(grid), dim3(256), 0, st, (const S*)img, n, h, w, mul, *out)
  const int dd = out->dtype;
  if (src_dtype == MYOLO_F32 && dd == MYOLO_F16) LAUNCH(float, half_t);
  else if (src_dtype == MYOLO_F32 && dd == MYOLO_F32) LAUNCH(float, float);
  else if (src_dtype == MYOLO_F16 && dd == MYOLO_F16) LAUNCH(half_t, half_t);
  else if (src_dtype == MYOLO_F16 && dd == MYOLO_F32) LAUNCH(half_t, float);
  else if (src_dtype == MYOLO_U8 && dd == MYOLO_F16) LAUNCH(uint8_t, half_t);
  else if (src_dtype == MYOLO_U8 && dd == MYOLO_F32) LAUNCH(uint8_t, float);
  else return MYOLO_EINVAL;
#undef LAUNCH
  MYOLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int myolo_pack_weights_mt(const int64_t* jobs, const int32_t* chunks, int nchunks, int chunk_elems, void* stream) {
  if (!jobs || !chunks || nchunks < 0) return MYOLO_EINVAL;
  if (nchunks == 0) return 0;
  if (chunk_elems <= 0) {                   // tiled mode: chunks = {job, tile of PACK_TILE (co, ci) pairs}; ntaps <= MYOLO_MAX_TAPS
    // chunk_elems = -T: no job has more than T taps (the LDS tile is sized for it: 18 KB for 3x3 instead of 51 KB); 0: up to MYOLO_MAX_TAPS
    const int maxt = chunk_elems < 0 ? -chunk_elems : MYOLO_MAX_TAPS;
    if (maxt > MYOLO_MAX_TAPS) return MYOLO_EINVAL;
    size_t smem = (size_t)PACK_TILE * maxt * sizeof(float);
    if (smem < (size_t)64 * 64 * sizeof(float)) smem = (size_t)64 * 64 * sizeof(float);      // (the 64 x 64 floats of a 1x1 tile)
    hipLaunchKernelGGL(pack_weights_tiled_kernel, dim3(nchunks), dim3(256), smem, (hipStream_t)stream, jobs, chunks);
    MYOLO_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(pack_weights_mt_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, jobs, chunks, chunk_elems);
  MYOLO_CHECK_LAUNCH();
  return 0;
}
