// Batched non-maximum suppression for the detect.py / test.py path, all images in one launch sequence, no host round trip
// until the kept rows are read back (reference utils/general.py:421-509 + torchvision.ops.nms, call site general.py:493).
//
//   nms_filter : obj > conf (general.py:430,446), cls *= obj (462), xywh->xyxy (465), best class / multi-label (468-473),
//                wave-aggregated compaction into per-image candidate lists
//   nms_rank   : descending-score rank of every candidate (ties broken by the original row for determinism); candidates are
//                scattered to their sorted slot, truncated to max_nms (487-488)
//   nms_scan   : one workgroup per image walks the sorted list in chunks of 64: the wave resolves the chunk's internal
//                suppressions with 64-bit lane masks, then all threads mark the later boxes the chunk's survivors suppress.
//                IoU is taken on the class-offset boxes (box + cls*max_wh, 491-492) with torchvision's formula
//                inter/(a+b-inter), strict '>'; stops once max_det boxes are kept (494-495).
#include "myolo_dev.h"

namespace {

__device__ __forceinline__ float ldp(const void* p, int64_t i, int dt) {
  return dt == MYOLO_F16 ? (float)((const half_t*)p)[i] : ((const float*)p)[i];
}

// class_mask: bit j set = class j passes the `classes=` filter of general.py:476-477 (0 = no filter)
__global__ __launch_bounds__(256) void nms_filter_kernel(const void* pred, int dt, int A, int no, float conf, int multi,
                                                         int cap, int* counts, float* cand, int* cand_idx, uint64_t class_mask) {
  const int b = blockIdx.y;
  const int nc = no - 5;
  // fp16 predictions (detect.py --half): the reference's `x[:, 5:] *= x[:, 4:5]`, `xywh2xyxy` and threshold compares run in the input
  // dtype (general.py:446-473) before torch.cat with `j.float()` promotes the rows to fp32 -- every such result is rounded to fp16
  const bool hf = dt == MYOLO_F16;
  auto rh = [hf](float v) { return hf ? (float)(half_t)v : v; };
  conf = rh(conf);
  const int lane = threadIdx.x & 63;
  // one atomic per WAVE and append (not per candidate: same-address atomics retire at ~110 ns each -- 30 k candidates of a
  // multi-label test.py call were 3 ms of nothing else): ballot, the first active lane reserves popcount slots, lanes take their rank
  auto append = [&](bool pass, float x1, float y1, float x2, float y2, float sc, int cls, int idx) {
    const unsigned long long m = __ballot(pass);
    if (!m) return;
    int base = 0;
    if (lane == __ffsll((long long)m) - 1) base = atomicAdd(counts + b, __popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1, 64);
    if (pass) {
      const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
      if (slot < cap) {
        float* c = cand + ((int64_t)b * cap + slot) * 6;
        c[0] = x1; c[1] = y1; c[2] = x2; c[3] = y2; c[4] = sc; c[5] = (float)cls;
        cand_idx[(int64_t)b * cap + slot] = idx;
      }
    }
  };
  const int stride = gridDim.x * blockDim.x;
  const int first = blockIdx.x * blockDim.x + threadIdx.x;
  for (int a0 = first - lane; a0 < A; a0 += stride) {              // (whole waves iterate together: the ballots need every lane)
    const int a = a0 + lane;
    const bool in = a < A;
    const int64_t row = ((int64_t)b * A + (in ? a : 0)) * no;
    const float obj = ldp(pred, row + 4, dt);
    const bool live = in && obj > conf;
    if (!__ballot(live)) continue;
    const float x = ldp(pred, row, dt), y = ldp(pred, row + 1, dt), w = ldp(pred, row + 2, dt), h = ldp(pred, row + 3, dt);
    const float hw = rh(w / 2), hh = rh(h / 2);
    const float x1 = rh(x - hw), y1 = rh(y - hh), x2 = rh(x + hw), y2 = rh(y + hh);
    if (multi && nc > 1) {
      for (int j = 0; j < nc; ++j) {
        const float s = rh(ldp(pred, row + 5 + j, dt) * obj);
        append(live && s > conf && (!class_mask || ((class_mask >> j) & 1ull)), x1, y1, x2, y2, s, j, a * nc + j);
      }
    } else {
      float best = -INFINITY; int bj = 0;
      for (int j = 0; j < nc; ++j) {
        const float s = rh(ldp(pred, row + 5 + j, dt) * obj);
        if (s > best) { best = s; bj = j; }                        // first maximum (torch.max)
      }
      append(live && best > conf && (!class_mask || ((class_mask >> bj) & 1ull)), x1, y1, x2, y2, best, bj, a);
    }
  }
}

__global__ __launch_bounds__(256) void nms_rank_kernel(const int* counts, const float* cand, const int* cand_idx, int cap,
                                                       int max_nms, float* sorted) {
  __shared__ float ss[256];
  __shared__ int si[256];
  const int b = blockIdx.y;
  int n = counts[b];
  if (n > cap) n = cap;
  const int i0 = blockIdx.x * 256;
  if (i0 >= n) return;
  const int i = i0 + threadIdx.x;
  const float* cb = cand + (int64_t)b * cap * 6;
  const int* ib = cand_idx + (int64_t)b * cap;
  const float s = i < n ? cb[(int64_t)i * 6 + 4] : 0.f;
  const int id = i < n ? ib[i] : 0;
  int rank = 0;
  for (int j0 = 0; j0 < n; j0 += 256) {
    const int j = j0 + threadIdx.x;
    ss[threadIdx.x] = j < n ? cb[(int64_t)j * 6 + 4] : -INFINITY;
    si[threadIdx.x] = j < n ? ib[j] : 0x7fffffff;
    __syncthreads();
    const int lim = n - j0 < 256 ? n - j0 : 256;
    for (int q = 0; q < lim; ++q) {
      const float sj = ss[q];
      rank += (sj > s || (sj == s && si[q] < id)) ? 1 : 0;
    }
    __syncthreads();
  }
  if (i < n && rank < max_nms) {
    float* d = sorted + ((int64_t)b * max_nms + rank) * 6;
#pragma unroll
    for (int q = 0; q < 6; ++q) d[q] = cb[(int64_t)i * 6 + q];
  }
}

// ---- descending-score order for LONG candidate lists (test.py: conf 0.001, multi_label -> up to A*nc = 322 560 candidates per image,
// 200 k typical): the rank-by-counting kernel above is O(n^2) (12 ms per image, 41 ms for a batch of 8).  Counting sort on the top
// 18 bits of the (positive) fp32 score, then an exact rank inside the bucket:
//   hist   : count per bucket (65 536 buckets per image: 9 mantissa bits per octave)
//   starts : descending exclusive scan -> first sorted slot of every bucket (one workgroup per image)
//   group  : every candidate takes a slot of its bucket in `order` (bucket-grouped candidate ids)
//   place  : exact rank = bucket start + #{members with a higher score, or the same score and a lower original row};
//            candidates ranked < max_nms are scattered into `sorted` (general.py:487-488)
constexpr int NB = 65536;
__device__ __forceinline__ int score_bucket(float s) { return (int)(__float_as_uint(s) >> 14); }     // 0 < s <= 1.0 -> < 0xFE01

__global__ __launch_bounds__(256) void nms_hist_kernel(const int* counts, const float* cand, int cap, int* hist) {
  const int b = blockIdx.y;
  int n = counts[b];
  if (n > cap) n = cap;
  const float* cb = cand + (int64_t)b * cap * 6;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    int k = score_bucket(cb[(int64_t)i * 6 + 4]);
    atomicAdd(hist + (int64_t)b * NB + (k < NB ? k : NB - 1), 1);
  }
}

__global__ __launch_bounds__(1024) void nms_starts_kernel(const int* hist, int* start) {
  __shared__ int part[1024];
  const int b = blockIdx.x, t = threadIdx.x;
  const int* h = hist + (int64_t)b * NB;
  // thread t owns buckets [hi - 63, hi], hi = NB - 1 - 64 t: the HIGHEST scores first
  const int hi = NB - 1 - 64 * t;
  int sum = 0;
  for (int q = 0; q < 64; ++q) sum += h[hi - q];
  part[t] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {                    // inclusive scan over the threads (Hillis-Steele)
    const int v = t >= o ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - sum;                                // candidates in all higher buckets
  int* st = start + (int64_t)b * NB;
  for (int q = 0; q < 64; ++q) { st[hi - q] = run; run += h[hi - q]; }
}

__global__ __launch_bounds__(256) void nms_group_kernel(const int* counts, const float* cand, int cap, const int* start, int* fill,
                                                        int* order) {
  const int b = blockIdx.y;
  int n = counts[b];
  if (n > cap) n = cap;
  const float* cb = cand + (int64_t)b * cap * 6;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    int k = score_bucket(cb[(int64_t)i * 6 + 4]);
    k = k < NB ? k : NB - 1;
    const int pos = start[(int64_t)b * NB + k] + atomicAdd(fill + (int64_t)b * NB + k, 1);
    order[(int64_t)b * cap + pos] = i;
  }
}

__global__ __launch_bounds__(256) void nms_place_kernel(const int* counts, const float* cand, const int* cand_idx, int cap,
                                                        const int* start, const int* hist, const int* order, int max_nms,
                                                        float* sorted) {
  const int b = blockIdx.y;
  int n = counts[b];
  if (n > cap) n = cap;
  const float* cb = cand + (int64_t)b * cap * 6;
  const int* ib = cand_idx + (int64_t)b * cap;
  const int* ob = order + (int64_t)b * cap;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float s = cb[(int64_t)i * 6 + 4];
    const int id = ib[i];
    int k = score_bucket(s);
    k = k < NB ? k : NB - 1;
    const int s0 = start[(int64_t)b * NB + k], m = hist[(int64_t)b * NB + k];
    if (s0 >= max_nms) continue;                          // the whole bucket lies beyond the truncation
    int rank = s0;
    for (int q = 0; q < m; ++q) {
      const int j = ob[s0 + q];
      const float sj = cb[(int64_t)j * 6 + 4];
      rank += (sj > s || (sj == s && ib[j] < id)) ? 1 : 0;
    }
    if (rank < max_nms) {
      float* d = sorted + ((int64_t)b * max_nms + rank) * 6;
#pragma unroll
      for (int q = 0; q < 6; ++q) d[q] = cb[(int64_t)i * 6 + q];
    }
  }
}

__device__ __forceinline__ bool iou_gt(float ax1, float ay1, float ax2, float ay2, float bx1, float by1, float bx2, float by2,
                                       float thr) {
  // disjoint boxes (in particular boxes of different classes: their offsets differ by >= max_wh) have IoU exactly 0:
  // four compares instead of the division.  Overlapping ones use torchvision's formula inter/(a+b-inter), strict '>'.
  const float w = fminf(ax2, bx2) - fmaxf(ax1, bx1), h = fminf(ay2, by2) - fmaxf(ay1, by1);
  if (w <= 0.f || h <= 0.f) return 0.f > thr;
  const float aa = (ax2 - ax1) * (ay2 - ay1), ab = (bx2 - bx1) * (by2 - by1);
  const float inter = w * h;
  return inter / (aa + ab - inter) > thr;
}

constexpr int SCAN_THREADS = 1024;
constexpr int MAXW = 1024;            // removed-bit words: up to 65536 sorted candidates per image
constexpr int LDS_BOXES = 6144;       // class-offset boxes cached in LDS (96 KB); later ones are re-read from L2

__global__ __launch_bounds__(SCAN_THREADS) void nms_scan_kernel(const int* counts, const float* sorted, int cap, int max_nms,
                                                                int max_det, float iou_thr, float max_wh, int agnostic,
                                                                float* out, int* nkeep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* removed = reinterpret_cast<unsigned long long*>(smem);              // [MAXW]
  unsigned long long* cmask = removed + MAXW;                                              // [64] intra-chunk masks
  f4_t* kbox = reinterpret_cast<f4_t*>(cmask + 64);                                        // [64] survivors of the chunk
  f4_t* lbox = kbox + 64;                                                                  // [LDS_BOXES] offset boxes
  __shared__ int s_nk, s_total;
  const int b = blockIdx.x;
  int m = counts[b];
  if (m > cap) m = cap;
  if (m > max_nms) m = max_nms;
  const float* sb = sorted + (int64_t)b * max_nms * 6;
  const int tid = threadIdx.x;
  const int words = (m + 63) >> 6;
  const float off = agnostic ? 0.f : max_wh;
  for (int i = tid; i < words; i += SCAN_THREADS) removed[i] = 0ull;
  for (int i = tid; i < m && i < LDS_BOXES; i += SCAN_THREADS) {
    const float o = sb[(int64_t)i * 6 + 5] * off;                     // class offset (general.py:491-492), fp32 like the reference
    lbox[i] = f4_t{sb[(int64_t)i * 6] + o, sb[(int64_t)i * 6 + 1] + o, sb[(int64_t)i * 6 + 2] + o, sb[(int64_t)i * 6 + 3] + o};
  }
  if (tid == 0) s_total = 0;
  __syncthreads();
  auto obox = [&](int i) -> f4_t {
    if (i < LDS_BOXES) return lbox[i];
    const float o = sb[(int64_t)i * 6 + 5] * off;
    return f4_t{sb[(int64_t)i * 6] + o, sb[(int64_t)i * 6 + 1] + o, sb[(int64_t)i * 6 + 2] + o, sb[(int64_t)i * 6 + 3] + o};
  };
  for (int c = 0; c < words; ++c) {
    const int base = c << 6;
    const int lim = m - base < 64 ? m - base : 64;
    const unsigned long long limmask = lim == 64 ? ~0ull : ((1ull << lim) - 1ull);
    if ((~removed[c] & limmask) == 0ull) continue;     // every box of the chunk is already suppressed (uniform: read after a barrier)
    if (tid < 64) cmask[tid] = 0ull;
    __syncthreads();
    // (a) the chunk's 64x64 suppression bits, all threads: pair (i, j > i)
    for (int pr = tid; pr < 64 * 64; pr += SCAN_THREADS) {
      const int i = pr >> 6, j = pr & 63;
      if (j > i && j < lim) {
        const f4_t a = obox(base + i), bb = obox(base + j);
        if (iou_gt(a[0], a[1], a[2], a[3], bb[0], bb[1], bb[2], bb[3], iou_thr)) atomicOr(&cmask[i], 1ull << j);
      }
    }
    __syncthreads();
    // (b) wave 0 resolves the chunk serially with lane-held masks
    if (tid < 64) {
      const unsigned long long mask = cmask[tid];
      unsigned long long rem = removed[c];
      unsigned long long keepbits = 0ull;
      for (int i2 = 0; i2 < lim; ++i2) {
        const unsigned int lo = __shfl((unsigned int)mask, i2, 64), hi = __shfl((unsigned int)(mask >> 32), i2, 64);
        if (!((rem >> i2) & 1ull)) {
          keepbits |= 1ull << i2;
          rem |= ((unsigned long long)hi << 32) | lo;
        }
      }
      const bool mine = (keepbits >> tid) & 1ull;
      const int pos = __popcll(keepbits & ((1ull << tid) - 1ull));
      const int total = s_total;
      if (mine) {
        kbox[pos] = obox(base + tid);
        const int o = total + pos;
        if (o < max_det) {
          float* d = out + ((int64_t)b * max_det + o) * 6;
#pragma unroll
          for (int q = 0; q < 6; ++q) d[q] = sb[(int64_t)(base + tid) * 6 + q];
        }
      }
      if (tid == 0) { s_nk = __popcll(keepbits); s_total = total + __popcll(keepbits); }
    }
    __syncthreads();
    const int nk = s_nk;
    if (s_total >= max_det) break;
    // (c) later boxes suppressed by this chunk's survivors
    for (int j = base + 64 + tid; j < m; j += SCAN_THREADS) {
      if ((removed[j >> 6] >> (j & 63)) & 1ull) continue;
      const f4_t bj = obox(j);
      bool dead = false;
      for (int q = 0; q < nk && !dead; ++q) {
        const f4_t kq = kbox[q];
        dead = iou_gt(kq[0], kq[1], kq[2], kq[3], bj[0], bj[1], bj[2], bj[3], iou_thr);
      }
      if (dead) atomicOr(&removed[j >> 6], 1ull << (j & 63));
    }
    __syncthreads();
  }
  if (tid == 0) nkeep[b] = s_total < max_det ? s_total : max_det;
}

}  // namespace

extern "C" int myolo_nms(const void* pred, int dtype, int batch, int A, int no, float conf_thres, float iou_thres,
                         int multi_label, int agnostic, float max_wh, int max_nms, int max_det, int cap, int32_t* counts,
                         float* cand, int32_t* cand_idx, float* sorted, float* out, int32_t* nkeep, uint64_t class_mask,
                         int32_t* sort_ws, void* stream) {
  if (class_mask && no - 5 > 64) return MYOLO_EINVAL;
  if (!pred || (dtype != MYOLO_F16 && dtype != MYOLO_F32) || batch < 1 || A < 1 || no < 6 || cap < 1 || max_det < 1 ||
      max_nms < 1 || max_nms > MAXW * 64 || !counts || !cand || !cand_idx || !sorted || !out || !nkeep)
    return MYOLO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(counts, 0, batch * sizeof(int32_t), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(nms_filter_kernel, dim3(grid_for(A, 256, 1024), batch), dim3(256), 0, st, pred, dtype, A, no, conf_thres,
                     multi_label, cap, counts, cand, cand_idx, class_mask);
  if (sort_ws) {             // long lists: counting sort; sort_ws = int32 [batch][3*65536 + cap]
    int* hist = sort_ws;
    int* fill = hist + (int64_t)batch * NB;
    int* start = fill + (int64_t)batch * NB;
    int* order = start + (int64_t)batch * NB;
    e = hipMemsetAsync(hist, 0, (size_t)2 * batch * NB * sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    const int gx = grid_for(cap, 256, 1024);
    hipLaunchKernelGGL(nms_hist_kernel, dim3(gx, batch), dim3(256), 0, st, counts, cand, cap, hist);
    hipLaunchKernelGGL(nms_starts_kernel, dim3(batch), dim3(1024), 0, st, hist, start);
    hipLaunchKernelGGL(nms_group_kernel, dim3(gx, batch), dim3(256), 0, st, counts, cand, cap, start, fill, order);
    hipLaunchKernelGGL(nms_place_kernel, dim3(gx, batch), dim3(256), 0, st, counts, cand, cand_idx, cap, start, hist, order, max_nms, sorted);
  } else {
    hipLaunchKernelGGL(nms_rank_kernel, dim3((cap + 255) / 256, batch), dim3(256), 0, st, counts, cand, cand_idx, cap, max_nms, sorted);
  }
  const int scan_smem = MAXW * 8 + 64 * 8 + 64 * 16 + LDS_BOXES * 16;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(nms_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, scan_smem);
    if (ea != hipSuccess) return (int)ea;
    attr_set = true;
  }
  hipLaunchKernelGGL(nms_scan_kernel, dim3(batch), dim3(SCAN_THREADS), scan_smem, st, counts, sorted, cap, max_nms, max_det, iou_thres,
                     max_wh, agnostic, out, nkeep);
  MYOLO_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// test.py:230-262: which predictions of ONE image count as true positives at each IoU threshold.  pred [n][6] = (x1,y1,x2,y2,conf,
// cls) in NMS order, labels [m][5] = (cls, x1,y1,x2,y2), both in native image space.  Per target class, predictions of that class
// are visited in order; a prediction whose best-IoU target (utils/general.py:388-410 box_iou, first maximum) exceeds iouv[0] and
// is not yet taken takes it and gets correct[j][k] = iou > iouv[k].  The reference walks this with one .item() sync per detection.
namespace {
__global__ __launch_bounds__(256) void match_kernel(const float* __restrict__ pred, int n, const float* __restrict__ labels, int m,
                                                    const float* __restrict__ iouv, int niou, uint8_t* __restrict__ correct,
                                                    float* best_iou, int* best_t, uint8_t* taken) {
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const float* p = pred + (int64_t)j * 6;
    const float x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3], cls = p[5];
    const float area1 = (x2 - x1) * (y2 - y1);
    float best = -1.f;
    int bt = -1;
    for (int t = 0; t < m; ++t) {
      const float* l = labels + (int64_t)t * 5;
      if (l[0] != cls) continue;
      const float area2 = (l[3] - l[1]) * (l[4] - l[2]);
      float iw = fminf(x2, l[3]) - fmaxf(x1, l[1]), ih = fminf(y2, l[4]) - fmaxf(y1, l[2]);
      iw = iw > 0.f ? iw : 0.f;
      ih = ih > 0.f ? ih : 0.f;
      const float inter = __fmul_rn(iw, ih);
      const float iou = __fdiv_rn(inter, __fsub_rn(__fadd_rn(area1, area2), inter));
      if (bt < 0 || iou > best) { best = iou; bt = t; }             // torch.max(1): first maximum
    }
    best_iou[j] = best;
    best_t[j] = bt;
    for (int k = 0; k < niou; ++k) correct[(int64_t)j * niou + k] = 0;
  }
  for (int t = threadIdx.x; t < m; t += blockDim.x) taken[t] = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float thr0 = iouv[0];
    for (int j = 0; j < n; ++j) {
      const int t = best_t[j];
      const float iou = best_iou[j];
      if (t < 0 || !(iou > thr0) || taken[t]) continue;
      taken[t] = 1;
      for (int k = 0; k < niou; ++k) correct[(int64_t)j * niou + k] = iou > iouv[k] ? 1 : 0;
    }
  }
}
}  // namespace

extern "C" int myolo_match_predictions(const float* pred, int n, const float* labels, int m, const float* iouv, int niou,
                                       uint8_t* correct, void* ws, int64_t ws_bytes, void* stream) {
  if (n < 0 || m < 0 || niou < 1 || !iouv || (n > 0 && (!pred || !correct)) || (m > 0 && !labels)) return MYOLO_EINVAL;
  if (n == 0) return 0;
  const int64_t need = (int64_t)n * 8 + ((m + 15) / 16) * 16;
  if (!ws || ws_bytes < need || ((uintptr_t)ws & 7)) return MYOLO_EINVAL;
  float* best_iou = reinterpret_cast<float*>(ws);
  int* best_t = reinterpret_cast<int*>(best_iou + n);
  uint8_t* taken = reinterpret_cast<uint8_t*>(best_t + n);
  hipLaunchKernelGGL(match_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, n, labels, m, iouv, niou, correct, best_iou, best_t,
                     taken);
  MYOLO_CHECK_LAUNCH();
  return 0;
}

