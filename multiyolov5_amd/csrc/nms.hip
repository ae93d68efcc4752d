// Batched non-maximum suppression for the detect.py / test.py path, all images in one launch sequence, no host round trip
// until the kept rows are read back (reference utils/general.py:421-509 + torchvision.ops.nms, call site general.py:493).
//
//   nms_filter : obj > conf (general.py:430,446), cls *= obj (462), xywh->xyxy (465), best class / multi-label (468-473),
//                wave-aggregated compaction into per-image candidate lists
//   nms_rank   : (lists longer than 8192 only) descending-score rank of every candidate (ties broken by the original row for
//                determinism); candidates are scattered to their sorted slot, truncated to max_nms (487-488)
//   nms_scan   : one workgroup per image; sorts short lists itself (bitonic network over 64-bit keys in LDS), then walks the sorted
//                list in chunks of 64 against the keep list (see the kernel).  IoU is taken on the class-offset boxes
//                (box + cls*max_wh, 491-492) with torchvision's formula inter/(a+b-inter), strict '>'; stops once max_det boxes
//                are kept (494-495).
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

constexpr int RANK_SKIP_BELOW = 8192;   // = SORT_MAX: lists this short are sorted in LDS by nms_scan_kernel
__global__ __launch_bounds__(256) void nms_rank_kernel(const int* counts, const float* cand, const int* cand_idx, int cap,
                                                       int max_nms, float* sorted, int lds_sort) {
  __shared__ float ss[256];
  __shared__ int si[256];
  const int b = blockIdx.y;
  int n = counts[b];
  if (n > cap) n = cap;
  const int i0 = blockIdx.x * 256;
  if (i0 >= n || (lds_sort && n <= RANK_SKIP_BELOW)) return;
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

__device__ __forceinline__ float box_area(const f4_t a) { return __fmul_rn(a[2] - a[0], a[3] - a[1]); }

// torchvision's test inter/(a+b-inter) > thr (strict), every operation individually rounded (no fused multiply-add: the areas are
// separate float products in the CPU kernel).  Disjoint boxes (in particular boxes of different classes: their offsets differ by
// >= max_wh) have IoU exactly 0: six min/max/sub and two compares; the division only runs when SOME lane of the wave overlaps.
__device__ __forceinline__ bool iou_gt(const f4_t a, float aa, const f4_t b, float ab, float thr) {
  const float w = fminf(a[2], b[2]) - fmaxf(a[0], b[0]), h = fminf(a[3], b[3]) - fmaxf(a[1], b[1]);
  const bool ov = w > 0.f && h > 0.f;
  bool r = !ov && (0.f > thr);
  if (__ballot(ov)) {
    const float inter = __fmul_rn(w, h);
    r = ov ? __fdiv_rn(inter, __fsub_rn(__fadd_rn(aa, ab), inter)) > thr : r;
  }
  return r;
}

constexpr int SCAN_THREADS = 1024;
constexpr int SCAN_WAVES = SCAN_THREADS / 64;
constexpr int MAX_SORTED = 65536;     // sorted candidates one image's scan can walk (max_nms is 30000, general.py:435)
constexpr int LDS_BOXES = 6144;       // class-offset boxes cached in LDS (96 KB); later ones are re-read from L2
constexpr int SORT_MAX = 8192;        // candidate lists up to this length are sorted in LDS by the scan kernel itself (64 KB of keys)
constexpr int SORT_PER_THREAD = SORT_MAX / SCAN_THREADS;
constexpr int KEPT_MAX = 1024;        // max_det (300, general.py:434) boxes of the keep list in LDS
constexpr int IDX_BITS = 19, SLOT_BITS = 13;     // LDS sort key: score (32) | inverted original row (19) | candidate slot (13)
constexpr int NBKT = 64;              // class buckets of the keep list (class id mod 64)
constexpr int NIL = 0xffff;

struct KeptBox { f4_t box; float area; int next; int pos; int pad; };     // 32 bytes: two ds_read_b128 per hop

// Greedy suppression of one image's score-sorted candidates (torchvision.ops.nms, call site general.py:493), one workgroup per image.
//  * short lists (detect.py: 1e3-1e4 candidates) are ordered HERE: 64-bit keys (score, original row, slot) go through a bitonic
//    network in LDS -- no rank launch, no O(n^2) pass (round 2: 0.2 ms), the keys' LDS is re-used by the box cache afterwards;
//  * the walk is LAZY: chunk c (64 boxes, one per lane) first learns which of its boxes the keep list so far suppresses, then the
//    chunk's own 64x64 triangle (wave w: rows 4w..4w+3, a ballot per row) -- no atomics, nothing is ever marked in LATER chunks, so the
//    work is bounded by n x kept pair tests and stops at max_det; wave 0 then resolves the chunk with scalar bit operations
//    (find-first-set over the live mask, readlane of the row words) and appends the survivors to the keep list.  Two barriers per
//    chunk, no global access inside the loop.  (Round 2 marked all later boxes per chunk through 64-bit LDS atomics: 1.15 ms for
//    7.5 k candidates; a single CU tests 64 pairs per ~13 cycles, so the pair count is what has to shrink:)
//  * the keep list is CHAINED per class bucket and wave (64 x 16 list heads): a lane only walks the kept boxes of its own class -- boxes
//    of different classes cannot overlap once the class offset is added, PROVIDED the un-offset coordinates span less than max_wh
//    (checked here over all candidates; otherwise every box goes to bucket 0 and the walk is exhaustive, like the reference's arithmetic).
__global__ __launch_bounds__(SCAN_THREADS) void nms_scan_kernel(const int* counts, const float* cand, const int* cand_idx,
                                                                float* sorted, int cap, int max_nms, int max_det, float iou_thr,
                                                                float max_wh, int agnostic, int lds_sort, float* out, int* nkeep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f4_t* lbox = reinterpret_cast<f4_t*>(smem);                                              // [LDS_BOXES] offset boxes (after the sort)
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);                  // [SORT_MAX]  (before)
  KeptBox* kept = reinterpret_cast<KeptBox*>(lbox + LDS_BOXES);                            // [KEPT_MAX] keep list
  unsigned long long* partial = reinterpret_cast<unsigned long long*>(kept + KEPT_MAX);    // [SCAN_WAVES]
  unsigned long long* cmask = partial + SCAN_WAVES;                                        // [64]
  unsigned short* khead = reinterpret_cast<unsigned short*>(cmask + 64);                   // [NBKT][SCAN_WAVES] chain heads
  float* srange = reinterpret_cast<float*>(khead + NBKT * SCAN_WAVES);                     // [2][SCAN_WAVES] coordinate min / max per wave
  unsigned char* lcls = reinterpret_cast<unsigned char*>(srange + 2 * SCAN_WAVES);         // [LDS_BOXES] class bucket of the cached boxes
  __shared__ int s_total;
  const int b = blockIdx.x;
  int m = counts[b];
  if (m > cap) m = cap;
  const bool sort_here = lds_sort && m <= SORT_MAX;
  const float* cb = cand + (int64_t)b * cap * 6;
  float* sb = sorted + (int64_t)b * max_nms * 6;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float off = agnostic ? 0.f : max_wh;
  if (tid == 0) s_total = 0;
  for (int i = tid; i < NBKT * SCAN_WAVES; i += SCAN_THREADS) khead[i] = (unsigned short)NIL;
  float cmin = INFINITY, cmax = -INFINITY;                                                  // range of the un-offset coordinates
  if (sort_here) {
    int npow = 64;
    while (npow < m) npow <<= 1;
    const int* ib = cand_idx + (int64_t)b * cap;
    for (int i = tid; i < npow; i += SCAN_THREADS) {
      unsigned long long k = 0ull;                                                          // padding sorts last (real scores are > 0)
      if (i < m)
        k = ((unsigned long long)__float_as_uint(cb[(int64_t)i * 6 + 4]) << 32) |
            ((unsigned long long)(((1u << IDX_BITS) - 1u) - (unsigned)ib[i]) << SLOT_BITS) | (unsigned)i;
      keys[i] = k;
    }
    __syncthreads();
    for (int k = 2; k <= npow; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < (npow >> 1); t += SCAN_THREADS) {
          const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i + j;
          const unsigned long long x = keys[i], y = keys[l];
          if ((x < y) == ((i & k) == 0)) { keys[i] = y; keys[l] = x; }                      // descending overall
        }
        __syncthreads();
      }
    }
    if (m > max_nms) m = max_nms;                                                           // general.py:487-488
    int slot[SORT_PER_THREAD];
#pragma unroll
    for (int q = 0; q < SORT_PER_THREAD; ++q) {
      const int p = tid + q * SCAN_THREADS;
      slot[q] = p < m ? (int)(keys[p] & ((1u << SLOT_BITS) - 1u)) : -1;
    }
    __syncthreads();                                                                        // the keys' LDS becomes the box cache
#pragma unroll
    for (int q = 0; q < SORT_PER_THREAD; ++q) {
      const int p = tid + q * SCAN_THREADS;
      if (slot[q] < 0) continue;
      const float* c = cb + (int64_t)slot[q] * 6;
      float r[6];
#pragma unroll
      for (int e = 0; e < 6; ++e) r[e] = c[e];
      float* d = sb + (int64_t)p * 6;                                                       // the sorted rows (output rows are copied from here)
#pragma unroll
      for (int e = 0; e < 6; ++e) d[e] = r[e];
      cmin = fminf(cmin, fminf(fminf(r[0], r[1]), fminf(r[2], r[3])));
      cmax = fmaxf(cmax, fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3])));
      const float o = r[5] * off;                                                           // class offset (general.py:491-492), fp32 like the reference
      if (p < LDS_BOXES) { lbox[p] = f4_t{r[0] + o, r[1] + o, r[2] + o, r[3] + o}; lcls[p] = (unsigned char)((int)r[5] & (NBKT - 1)); }
    }
  } else {
    if (m > max_nms) m = max_nms;
    for (int i = tid; i < m; i += SCAN_THREADS) {
      const float* r = sb + (int64_t)i * 6;
      cmin = fminf(cmin, fminf(fminf(r[0], r[1]), fminf(r[2], r[3])));
      cmax = fmaxf(cmax, fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3])));
      const float o = r[5] * off;
      if (i < LDS_BOXES) { lbox[i] = f4_t{r[0] + o, r[1] + o, r[2] + o, r[3] + o}; lcls[i] = (unsigned char)((int)r[5] & (NBKT - 1)); }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { cmin = fminf(cmin, __shfl_xor(cmin, o, 64)); cmax = fmaxf(cmax, __shfl_xor(cmax, o, 64)); }
  if (lane == 0) { srange[wv] = cmin; srange[SCAN_WAVES + wv] = cmax; }
  __threadfence_block();
  __syncthreads();
#pragma unroll
  for (int q = 0; q < SCAN_WAVES; ++q) { cmin = fminf(cmin, srange[q]); cmax = fmaxf(cmax, srange[SCAN_WAVES + q]); }
  // classes are disjoint after the offset iff every class's coordinate range [c*max_wh + cmin, c*max_wh + cmax] ends before the next
  // one begins (NaN coordinates fail the compare: exhaustive walk)
  const bool by_class = !agnostic && (cmax - cmin < max_wh);
  auto obox = [&](int i) -> f4_t {
    if (i < LDS_BOXES) return lbox[i];
    const volatile float* r = sb + (int64_t)i * 6;                                          // (rows this workgroup may have written itself)
    const float o = r[5] * off;
    return f4_t{r[0] + o, r[1] + o, r[2] + o, r[3] + o};
  };
  const int words = (m + 63) >> 6;
  int total = 0;
  for (int c = 0; c < words; ++c) {
    const int base = c << 6;
    const int lim = m - base < 64 ? m - base : 64;
    const unsigned long long limmask = lim == 64 ? ~0ull : ((1ull << lim) - 1ull);
    const bool valid = lane < lim;
    const f4_t bj = valid ? obox(base + lane) : f4_t{0.f, 0.f, 0.f, 0.f};
    const float aj = box_area(bj);
    int bkt = 0;
    if (by_class && valid)
      bkt = base + lane < LDS_BOXES ? (int)lcls[base + lane] : (((int)*(const volatile float*)(sb + (int64_t)(base + lane) * 6 + 5)) & (NBKT - 1));
    // (a) this chunk against the keep list so far: wave wv walks chain (bucket, wv); one ballot = the wave's removed word
    bool dead = false;
    int k = valid ? khead[bkt * SCAN_WAVES + wv] : NIL;
    while (__ballot(k != NIL)) {
      const bool act = k != NIL;
      const KeptBox kb = kept[act ? k : 0];
      const bool hit = iou_gt(kb.box, kb.area, bj, aj, iou_thr);
      dead = dead || (act && hit);
      k = act ? kb.next : NIL;
    }
    const unsigned long long dw = __ballot(dead && valid);
    if (lane == 0) partial[wv] = dw;
    // (b) the chunk's own upper triangle: row i suppresses column j > i
#pragma unroll
    for (int r = 0; r < 64 / SCAN_WAVES; ++r) {
      const int i = wv * (64 / SCAN_WAVES) + r;
      unsigned long long rw = 0ull;
      if (i < lim) {
        const f4_t bi = obox(base + i);
        rw = __ballot(iou_gt(bi, box_area(bi), bj, aj, iou_thr) && valid && lane > i);
      }
      if (lane == 0) cmask[i] = rw;
    }
    __syncthreads();
    if (wv == 0) {
      unsigned long long remv = ~limmask;
#pragma unroll
      for (int q = 0; q < SCAN_WAVES; ++q) remv |= partial[q];
      const unsigned long long rowv = cmask[lane];
      const unsigned int row_lo = (unsigned int)rowv, row_hi = (unsigned int)(rowv >> 32);
      // wave-uniform from here: scalar registers
      // (the builtins return int: without the unsigned casts a set bit 31 of the low word would sign-extend over the high word)
      unsigned long long rem = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((unsigned int)(remv >> 32)) << 32) |
                               (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((unsigned int)remv);
      unsigned long long keepbits = 0ull;
      int budget = max_det - total;
      while (~rem != 0ull && budget > 0) {
        const int i = __ffsll((long long)~rem) - 1;
        keepbits |= 1ull << i;
        --budget;
        const unsigned long long rw = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane(row_hi, i) << 32) |
                                      (unsigned long long)(unsigned int)__builtin_amdgcn_readlane(row_lo, i);
        rem |= rw | (1ull << i);
      }
      const bool mine = (keepbits >> lane) & 1ull;
      const int rel = __popcll(keepbits & ((1ull << lane) - 1ull));
      const int pos = total + rel;
      const int nk = __popcll(keepbits);
      // survivors join their chain (bucket, pos % 16); 16 at a time, so that no two lanes of a round share a chain head
      for (int r0 = 0; r0 < nk; r0 += SCAN_WAVES) {
        if (mine && rel >= r0 && rel < r0 + SCAN_WAVES) {
          const int ch = bkt * SCAN_WAVES + (pos & (SCAN_WAVES - 1));
          KeptBox kb;
          kb.box = bj; kb.area = aj; kb.next = khead[ch]; kb.pos = base + lane; kb.pad = 0;
          kept[pos] = kb;
          khead[ch] = (unsigned short)pos;
        }
      }
      if (lane == 0) s_total = total + nk;
    }
    __syncthreads();
    total = s_total;
    if (total >= max_det) break;
  }
  // the kept rows, in keep order (= descending score): x1, y1, x2, y2, conf, cls as the filter wrote them
  for (int e = tid; e < total * 6; e += SCAN_THREADS) {
    const int k = e / 6, q = e - k * 6;
    out[((int64_t)b * max_det + k) * 6 + q] = *(const volatile float*)(sb + (int64_t)kept[k].pos * 6 + q);
  }
  if (tid == 0) nkeep[b] = total;
}

}  // namespace

extern "C" int myolo_nms(const void* pred, int dtype, int batch, int A, int no, float conf_thres, float iou_thres,
                         int multi_label, int agnostic, float max_wh, int max_nms, int max_det, int cap, int32_t* counts,
                         float* cand, int32_t* cand_idx, float* sorted, float* out, int32_t* nkeep, uint64_t class_mask,
                         int32_t* sort_ws, void* stream) {
  if (class_mask && no - 5 > 64) return MYOLO_EINVAL;
  if (!pred || (dtype != MYOLO_F16 && dtype != MYOLO_F32) || batch < 1 || A < 1 || no < 6 || cap < 1 || max_det < 1 ||
      max_det > KEPT_MAX || max_nms < 1 || max_nms > MAX_SORTED || !counts || !cand || !cand_idx || !sorted || !out || !nkeep)
    return MYOLO_EINVAL;
  static_assert(RANK_SKIP_BELOW == SORT_MAX && SORT_MAX == (1 << SLOT_BITS), "LDS sort key layout");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(counts, 0, batch * sizeof(int32_t), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(nms_filter_kernel, dim3(grid_for(A, 256, 1024), batch), dim3(256), 0, st, pred, dtype, A, no, conf_thres,
                     multi_label, cap, counts, cand, cand_idx, class_mask);
  // short single-label lists are ordered inside the scan kernel (original rows must fit the key's 19 bits)
  const int lds_sort = (!sort_ws && (int64_t)cap <= (1ll << IDX_BITS)) ? 1 : 0;
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
    hipLaunchKernelGGL(nms_rank_kernel, dim3((cap + 255) / 256, batch), dim3(256), 0, st, counts, cand, cand_idx, cap, max_nms, sorted,
                       lds_sort);
  }
  const int scan_smem = LDS_BOXES * 16 + KEPT_MAX * (int)sizeof(KeptBox) + SCAN_WAVES * 8 + 64 * 8 + NBKT * SCAN_WAVES * 2 + 2 * SCAN_WAVES * 4 + LDS_BOXES;
  static_assert(SORT_MAX * 8 <= LDS_BOXES * 16, "the sort keys live in the box cache's LDS");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(nms_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, scan_smem);
    if (ea != hipSuccess) return (int)ea;
    attr_set = true;
  }
  hipLaunchKernelGGL(nms_scan_kernel, dim3(batch), dim3(SCAN_THREADS), scan_smem, st, counts, cand, cand_idx, sorted, cap, max_nms,
                     max_det, iou_thres, max_wh, agnostic, lds_sort, out, nkeep);
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

