// Batched non-maximum suppression for the detect.py / test.py path, all images in one launch sequence, no host round trip
// until the kept rows are read back (reference utils/general.py:421-509 + torchvision.ops.nms, call site general.py:493).
//
//   nms_filter : obj > conf (general.py:430,446), cls *= obj (462), xywh->xyxy (465), best class / multi-label (468-473),
//                wave-aggregated compaction into per-image candidate lists
//   nms_rank   : (lists longer than 8192 only) descending-score rank of every candidate (ties broken by the original row for
//                determinism); candidates are scattered to their sorted slot, truncated to max_nms (487-488)
//   (short lists: rank2 / scatter / mask / mscan / merge, class by class on the whole device -- see "SHORT lists" below)
//   nms_scan   : one workgroup per image; sorts short lists itself (bitonic network over 64-bit keys in LDS), then walks the sorted
//                list in chunks of 64 against the keep list (see the kernel).  IoU is taken on the class-offset boxes
//                (box + cls*max_wh, 491-492) with torchvision's formula inter/(a+b-inter), strict '>'; stops once max_det boxes
//                are kept (494-495).
#include "myolo_dev.h"

namespace {

// ---- per-image side data of the matrix path (short lists): written by the filter, the segment table by nms_rank2's first workgroup ----
constexpr int NMS_MAXC = 128;                 // classes the segment table distinguishes (class byte of the key & 127 .. nc <= 128 checked on the host)
constexpr int NMS_COPIES = 32;                // the filter's workgroups add into copy (workgroup % 32): same-address atomics retire at ~110 ns each,
                                              // 504 workgroups on ONE counter per class made the filter 51 us instead of 11
constexpr int NMS_AUX_HIST = 0;               // [NMS_COPIES][NMS_MAXC] candidates per class
constexpr int NMS_AUX_HI = NMS_COPIES * NMS_MAXC;     // [NMS_COPIES] max(x2, y2) over the candidates, order-preserving unsigned encoding (0 = -inf)
constexpr int NMS_AUX_NLO = NMS_AUX_HI + NMS_COPIES;  // [NMS_COPIES] max(-x1, -y1)
constexpr int NMS_AUX_KCOUNT = NMS_AUX_NLO + NMS_COPIES;      // boxes kept so far (all segments)
constexpr int NMS_AUX_OK = NMS_AUX_KCOUNT + 1;        // 1: the matrix path owns this image (else the lazy scan kernel does)
constexpr int NMS_AUX_NSEG = NMS_AUX_KCOUNT + 2;
constexpr int NMS_AUX_SEG = NMS_AUX_KCOUNT + 8;       // [NMS_MAXC][2] segment s: first sorted slot (64-aligned), candidates
constexpr int NMS_AUX_CSEG = NMS_AUX_SEG + 2 * NMS_MAXC;    // [NMS_MAXC][2] class c: its segment, candidates of the classes sorted before it
constexpr int NMS_AUX_INTS = NMS_AUX_CSEG + 2 * NMS_MAXC;
__device__ __forceinline__ unsigned int ford(float f) {           // monotonic float -> unsigned
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float funord(unsigned int u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__device__ __forceinline__ float ldp(const void* p, int64_t i, int dt) {
  return dt == MYOLO_F16 ? (float)((const half_t*)p)[i] : ((const float*)p)[i];
}

// class_mask: bit j set = class j passes the `classes=` filter of general.py:476-477 (0 = no filter)
__global__ __launch_bounds__(256) void nms_filter_kernel(const void* pred, int dt, int A, int no, float conf, int multi,
                                                         int cap, int* counts, float* cand, int* cand_idx, uint64_t class_mask,
                                                         unsigned long long* keys, int* rank, int64_t pred_bytes, int* aux, int keymode) {
  const int b = blockIdx.y;
  const int nc = no - 5;
  // fp16 predictions (detect.py --half): the reference's `x[:, 5:] *= x[:, 4:5]`, `xywh2xyxy` and threshold compares run in the input
  // dtype (general.py:446-473) before torch.cat with `j.float()` promotes the rows to fp32 -- every such result is rounded to fp16
  const bool hf = dt == MYOLO_F16;
  auto rh = [hf](float v) { return hf ? (float)(half_t)v : v; };
  conf = rh(conf);
  const int lane = threadIdx.x & 63;
  // one atomic per WORKGROUP and append (same-address atomics retire at ~18-110 ns each whatever else the chip does: one per
  // candidate made a multi-label test.py call 3 ms of nothing else, one per wave still 2016 x 18 ns = 36 us for the 129 k rows of a
  // 2048x1024 frame): wave ballots meet in LDS, thread 0 reserves the workgroup's slots, every lane takes its rank.  Called uniformly by
  // all four waves (two barriers per call; the LDS cells alternate between calls so a fast wave cannot overwrite a slow wave's input).
  __shared__ int s_wc[2][4], s_base[2];
  // matrix path (keys != nullptr): per-class candidate counts and the range of the un-offset coordinates of this image (NmsAux)
  __shared__ int s_hist[NMS_MAXC];
  if (aux) {
    for (int c = threadIdx.x; c < NMS_MAXC; c += 256) s_hist[c] = 0;
    __syncthreads();
  }
  float t_hi = -INFINITY, t_nlo = -INFINITY;
  int call_parity = 0;
  auto append = [&](bool pass, float x1, float y1, float x2, float y2, float sc, int cls, int idx) {
    const unsigned long long m = __ballot(pass);
    const int wave = threadIdx.x >> 6, par = call_parity;
    call_parity ^= 1;
    if (lane == 0) s_wc[par][wave] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
      const int tot = s_wc[par][0] + s_wc[par][1] + s_wc[par][2] + s_wc[par][3];
      s_base[par] = tot ? atomicAdd(counts + b, tot) : 0;
    }
    __syncthreads();
    int base = s_base[par];
    for (int w = 0; w < wave; ++w) base += s_wc[par][w];
    if (pass) {
      const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
      if (slot < cap) {
        float* c = cand + ((int64_t)b * cap + slot) * 6;
        c[0] = x1; c[1] = y1; c[2] = x2; c[3] = y2; c[4] = sc; c[5] = (float)cls;
        cand_idx[(int64_t)b * cap + slot] = idx;
        // descending order = descending key: score bits (positive floats order like integers), ties broken by the LOWER original row;
        // keymode 1: the class on top (class-major order: classes are suppressed independently, see nms_segs below)
        if (keys)
          keys[(int64_t)b * cap + slot] = keymode
              ? ((unsigned long long)(unsigned)cls << 56) | ((unsigned long long)__float_as_uint(sc) << 24) | (unsigned long long)(0xffffffu - (unsigned)idx)
              : ((unsigned long long)__float_as_uint(sc) << 32) | (0xffffffffu - (unsigned)idx);
        if (aux) {
          atomicAdd(&s_hist[cls & (NMS_MAXC - 1)], 1);
          t_hi = fmaxf(t_hi, fmaxf(x2, y2));
          t_nlo = fmaxf(t_nlo, fmaxf(-x1, -y1));
        }
      }
    }
  };
  // a workgroup takes 256 consecutive rows (256 * no elements, contiguous) per pass: 16-byte loads into LDS, then every thread reads its
  // own row from there (a row is 15 halves = 30 bytes: per-thread element loads were 2-byte accesses 30 bytes apart, 29 us for 129 k rows)
  extern __shared__ __attribute__((aligned(16))) char frow[];
  const int es = hf ? 2 : 4;
  const int64_t img = (int64_t)b * A * no;
  for (int r0 = blockIdx.x * 256; r0 < A; r0 += gridDim.x * 256) {
    const int nrow = A - r0 < 256 ? A - r0 : 256;
    const int64_t byte0 = (img + (int64_t)r0 * no) * es;            // first byte of the pass; the prediction base is 16-byte aligned (host)
    const int64_t nbytes = (int64_t)nrow * no * es;
    const int64_t lo = byte0 & ~15ll, hi = (byte0 + nbytes + 15) & ~15ll;       // whole 16-byte vectors; nothing past the tensor's last byte is read
    const int shift = (int)(byte0 - lo);
    __syncthreads();
    for (int64_t v = lo + (int64_t)threadIdx.x * 16; v < hi; v += 256 * 16) {
      uint4 q = uint4{0u, 0u, 0u, 0u};
      if (v + 16 <= pred_bytes) q = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(pred) + v);
      else for (int k = 0; k < 16; ++k) if (v + k < pred_bytes) reinterpret_cast<char*>(&q)[k] = reinterpret_cast<const char*>(pred)[v + k];
      *reinterpret_cast<uint4*>(frow + (v - lo)) = q;
    }
    __syncthreads();
    const int a = r0 + threadIdx.x;
    const bool in = threadIdx.x < nrow;
    if (rank && in) rank[(int64_t)b * cap + a] = 0;                 // (single-label lists: cap == A, every slot is some row's index)
    const char* rowp = frow + shift + (int64_t)(in ? threadIdx.x : 0) * no * es;
    auto ld = [&](int k) -> float { return hf ? (float)reinterpret_cast<const half_t*>(rowp)[k] : reinterpret_cast<const float*>(rowp)[k]; };
    const float obj = ld(4);
    const bool live = in && obj > conf;
    const float x = ld(0), y = ld(1), w = ld(2), h = ld(3);
    const float hw = rh(w / 2), hh = rh(h / 2);
    const float x1 = rh(x - hw), y1 = rh(y - hh), x2 = rh(x + hw), y2 = rh(y + hh);
    if (multi && nc > 1) {
      for (int j = 0; j < nc; ++j) {
        const float s = rh(ld(5 + j) * obj);
        append(live && s > conf && (!class_mask || ((class_mask >> j) & 1ull)), x1, y1, x2, y2, s, j, a * nc + j);
      }
    } else {
      float best = -INFINITY; int bj = 0;
      for (int j = 0; j < nc; ++j) {
        const float s = rh(ld(5 + j) * obj);
        if (s > best) { best = s; bj = j; }                        // first maximum (torch.max)
      }
      append(live && best > conf && (!class_mask || ((class_mask >> bj) & 1ull)), x1, y1, x2, y2, best, bj, a);
    }
  }
  if (aux) {
    __syncthreads();
    int* ax = aux + (int64_t)b * NMS_AUX_INTS;
    const int copy = blockIdx.x % NMS_COPIES;
    for (int c = threadIdx.x; c < NMS_MAXC; c += 256)
      if (s_hist[c]) atomicAdd(ax + NMS_AUX_HIST + copy * NMS_MAXC + c, s_hist[c]);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { t_hi = fmaxf(t_hi, __shfl_xor(t_hi, o, 64)); t_nlo = fmaxf(t_nlo, __shfl_xor(t_nlo, o, 64)); }
    __shared__ float s_rng[2][4];
    if (lane == 0) { s_rng[0][threadIdx.x >> 6] = t_hi; s_rng[1][threadIdx.x >> 6] = t_nlo; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const float hi = fmaxf(fmaxf(s_rng[0][0], s_rng[0][1]), fmaxf(s_rng[0][2], s_rng[0][3]));
      const float nlo = fmaxf(fmaxf(s_rng[1][0], s_rng[1][1]), fmaxf(s_rng[1][2], s_rng[1][3]));
      if (hi > -INFINITY) {
        atomicMax(reinterpret_cast<unsigned int*>(ax + NMS_AUX_HI + copy), ford(hi));
        atomicMax(reinterpret_cast<unsigned int*>(ax + NMS_AUX_NLO + copy), ford(nlo));
      }
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
  const float w = fminf(a[2], b[2]) - fmaxf(a[0], b[0]);
  if (!__ballot(w > 0.f)) return 0.f > thr;         // no lane overlaps along x (other classes sit >= max_wh away): done after 3 operations
  const float h = fminf(a[3], b[3]) - fmaxf(a[1], b[1]);
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
                                                                float max_wh, int agnostic, int lds_sort, float* out, int* nkeep, int dbg, const int* aux) {
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
  if (aux && aux[(int64_t)b * NMS_AUX_INTS + NMS_AUX_OK]) return;   // (the matrix path has this image)
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
  const int words = (dbg & 1) ? 0 : (m + 63) >> 6;      // dbg: profiling only (1 sort only, 2 no keep-list walk, 4 no triangle, 8 no resolve)
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
    int k = (valid && !(dbg & 2)) ? khead[bkt * SCAN_WAVES + wv] : NIL;
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
      if (i < lim && !(dbg & 4)) {
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
      if (dbg & 8) rem = ~0ull;
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


// ---- SHORT lists (detect.py: <= 8192 candidates): the pair tests leave the sequential walk altogether ---------------------------
//   rank2   : descending rank of every candidate, the whole device: [i block] x [j slice] workgroups count keys greater than their own
//             (keys are unique), partial counts meet in one atomicAdd per (candidate, slice).  Its first workgroup per image also builds
//             the SEGMENT TABLE: with the class on top of the key the order is class-major, and classes are suppressed independently
//             of each other -- box + cls*max_wh (general.py:491-492) cannot overlap across classes as long as the un-offset
//             coordinates span less than max_wh (checked over all candidates; otherwise, and for lists whose padded length does not
//             fit, the image is left to the lazy scan kernel).  A segment = one class, its sorted slots start on a multiple of 64
//   scatter : candidates to their (segment-aligned) sorted slot + the class-offset boxes + the score part of the key
//   mask    : the suppression bit matrix, upper triangle, one 64 x 64 tile per wave and pass, ONLY tiles inside a segment: with ten classes
//             a tenth of the n^2/2 pair tests (7.5 k candidates: 46 -> ~8 us)
//   mscan   : ONE wave per (image, segment) walks the segment's chunks with bit operations only and no barrier (see below); the
//             segments of an image run side by side (one 7.5 k list: 117 sequential chunks at 0.65 us -> ~12 per class); each appends its
//             kept slots and their score keys to the image's kept list
//   merge   : the kept boxes of all classes back into descending-score order (rank by counting among the kept), the first max_det of
//             them are the output (general.py:494-495) -- a class cannot contribute more than max_det, so each scan stops there
//   Chunk c of a scan needs (1) the OR over all kept rows of their word c: every lane gathers the words of up to MS_KR kept rows, issued
//   MS_P chunks ahead for the keep list as it stood then; (2) the contribution of boxes kept during the last MS_P chunks: those chunks'
//   own rows were loaded as a BAND of MS_P+1 words (diagonal + the next MS_P), so a survivor's later words are readlane'd into MS_P scalar
//   accumulators the moment it is kept; (3) the diagonal words of its own 64 rows (the band's first word).  All loads are
//   address-predictable MS_P chunks ahead: the dependent chain per chunk is a cross-lane OR, a find-first-set loop and a few readlanes.
constexpr int MS_P = 4;
constexpr int MS_KR = 5;                       // kept rows gathered per lane: max_det <= 64 * MS_KR = 320 (general.py:434 uses 300)
constexpr int MS_RS = SORT_MAX / 64 + 8;       // mask row stride in 64-bit words (the band may read MS_P words past the last chunk)
constexpr unsigned long long KEY56 = 0x00ffffffffffffffull;      // score | inverted row: the key without its class byte

__global__ __launch_bounds__(256) void nms_rank2_kernel(const int* counts, const unsigned long long* keys, int cap, int* rank, int* aux,
                                                        int keymode, float max_wh) {
  __shared__ unsigned long long sk[256];
  const int b = blockIdx.z;
  int n = counts[b];
  if (n > cap) n = cap;
  if (blockIdx.x == 0 && blockIdx.y == 0) {                              // the image's segment table (read by the kernels behind this one)
    int* ax = aux + (int64_t)b * NMS_AUX_INTS;
    __shared__ int sh[NMS_MAXC];
    __shared__ unsigned int srng[2];
    if (threadIdx.x < NMS_MAXC) {
      int h = 0;
#pragma unroll 8
      for (int k = 0; k < NMS_COPIES; ++k) h += ax[NMS_AUX_HIST + k * NMS_MAXC + threadIdx.x];
      sh[threadIdx.x] = h;
    } else if (threadIdx.x < NMS_MAXC + 2) {
      unsigned int m = 0;
      for (int k = 0; k < NMS_COPIES; ++k) { const unsigned int v = (unsigned int)ax[(threadIdx.x == NMS_MAXC ? NMS_AUX_HI : NMS_AUX_NLO) + k]; m = v > m ? v : m; }
      srng[threadIdx.x - NMS_MAXC] = m;
    }
    __syncthreads();
    if (threadIdx.x < NMS_MAXC) {
      // descending key order = descending class: class c sits behind every class above it (each thread sums the classes above its own)
      const int c = threadIdx.x, h = sh[c];
      int before = 0, slot = 0, sg = 0;
      for (int c2 = c + 1; c2 < NMS_MAXC; ++c2) { const int h2 = sh[c2]; before += h2; slot += (h2 + 63) & ~63; sg += h2 > 0 ? 1 : 0; }
      if (keymode && h > 0) {
        ax[NMS_AUX_SEG + 2 * sg] = slot; ax[NMS_AUX_SEG + 2 * sg + 1] = h;
        ax[NMS_AUX_CSEG + 2 * c] = sg; ax[NMS_AUX_CSEG + 2 * c + 1] = before;
      }
      if (c == 0) {
        const float span = funord(srng[0]) + funord(srng[1]);             // max(x2,y2) - min(x1,y1) over the candidates
        bool ok = n <= SORT_MAX;
        int nseg = 1;
        if (keymode && n > 0) {
          nseg = sg + (h > 0 ? 1 : 0);
          ok = ok && span < max_wh && slot + ((h + 63) & ~63) <= SORT_MAX && before + h == n;
        } else if (ok) {                                                  // one segment: plain score order (agnostic / keys without class /
          ax[NMS_AUX_SEG] = 0; ax[NMS_AUX_SEG + 1] = n;                   // empty list); class -> segment 0, nothing before: the zeroed table
        }
        ax[NMS_AUX_NSEG] = ok ? nseg : 0;
        ax[NMS_AUX_OK] = ok ? 1 : 0;
      }
    }
  }
  if (n > SORT_MAX || (int)blockIdx.x * 256 >= n) return;
  const int tid = threadIdx.x, i = blockIdx.x * 256 + tid;
  const unsigned long long* kb = keys + (int64_t)b * cap;
  const unsigned long long ki = i < n ? kb[i] : ~0ull;
  const int per = (n + (int)gridDim.y - 1) / (int)gridDim.y;
  const int j0 = blockIdx.y * per, j1 = j0 + per < n ? j0 + per : n;
  int cnt = 0;
  for (int jt = j0; jt < j1; jt += 256) {
    sk[tid] = jt + tid < j1 ? kb[jt + tid] : 0ull;
    __syncthreads();
    const int lim = j1 - jt < 256 ? j1 - jt : 256;
    for (int q = 0; q < lim; ++q) cnt += sk[q] > ki ? 1 : 0;
    __syncthreads();
  }
  if (i < n && cnt) atomicAdd(rank + (int64_t)b * cap + i, cnt);
}

__global__ __launch_bounds__(256) void nms_scatter_kernel(const int* counts, const float* cand, const int* rank, const unsigned long long* keys,
                                                          int cap, int max_nms, float max_wh, int agnostic, int keymode, const int* aux,
                                                          float* sorted, f4_t* obx, unsigned long long* skey) {
  const int b = blockIdx.y;
  const int* ax = aux + (int64_t)b * NMS_AUX_INTS;
  if (!ax[NMS_AUX_OK]) return;
  int n = counts[b];
  if (n > cap) n = cap;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* c = cand + ((int64_t)b * cap + i) * 6;
  float v[6];
#pragma unroll
  for (int e = 0; e < 6; ++e) v[e] = c[e];
  const int cls = (int)v[5] & (NMS_MAXC - 1);
  const int sg = keymode ? ax[NMS_AUX_CSEG + 2 * cls] : 0, before = keymode ? ax[NMS_AUX_CSEG + 2 * cls + 1] : 0;
  const int r = ax[NMS_AUX_SEG + 2 * sg] + (rank[(int64_t)b * cap + i] - before);        // slot: segment start + rank inside the class
  if (r >= max_nms || r >= SORT_MAX) return;
  float* d = sorted + ((int64_t)b * max_nms + r) * 6;
#pragma unroll
  for (int e = 0; e < 6; ++e) d[e] = v[e];
  const float o = v[5] * (agnostic ? 0.f : max_wh);               // class offset (general.py:491-492), fp32 like the reference
  obx[(int64_t)b * SORT_MAX + r] = f4_t{v[0] + o, v[1] + o, v[2] + o, v[3] + o};
  const unsigned long long k = keys[(int64_t)b * cap + i];
  skey[(int64_t)b * SORT_MAX + r] = keymode ? (k & KEY56) : k;
}

__device__ __forceinline__ float rl_f(float v, int i) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i)); }
__device__ __forceinline__ unsigned long long rl_u64(unsigned long long v, int i) {
  return ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), i) << 32) |
         (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, i);
}

__global__ __launch_bounds__(256) void nms_mask_kernel(const int* aux, const f4_t* obx, unsigned long long* mask, float thr) {
  __shared__ short wseg[MS_RS];                                     // segment of every 64-slot word (-1: past the last segment)
  __shared__ int send[NMS_MAXC];                                    // one past the last candidate slot of every segment
  const int b = blockIdx.y;
  const int* ax = aux + (int64_t)b * NMS_AUX_INTS;
  if (!ax[NMS_AUX_OK]) return;
  const int nseg = ax[NMS_AUX_NSEG];
  for (int w = threadIdx.x; w < MS_RS; w += 256) wseg[w] = -1;
  __syncthreads();
  for (int sg = threadIdx.x; sg < nseg; sg += 256) {
    const int s0 = ax[NMS_AUX_SEG + 2 * sg], h = ax[NMS_AUX_SEG + 2 * sg + 1];
    send[sg] = s0 + h;
    for (int w = s0 >> 6; w < (s0 + h + 63) >> 6; ++w) wseg[w] = (short)sg;
  }
  __syncthreads();
  const int lastseg = nseg - 1;
  const int words = nseg > 0 ? (ax[NMS_AUX_SEG + 2 * lastseg] + ax[NMS_AUX_SEG + 2 * lastseg + 1] + 63) >> 6 : 0;
  const int lane = threadIdx.x & 63;
  const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), nw = gridDim.x * 4;
  const f4_t* ob = obx + (int64_t)b * SORT_MAX;
  unsigned long long* M = mask + (int64_t)b * SORT_MAX * MS_RS;
  const f4_t zero = f4_t{0.f, 0.f, 0.f, 0.f};
  for (int t = gw; t < words * words; t += nw) {
    const int rc = t / words, cc = t - rc * words;
    if (cc < rc || wseg[rc] != wseg[cc]) continue;                  // upper triangle (later boxes only), inside one segment
    const int m = send[wseg[rc]];
    const int row = rc * 64 + lane, col = cc * 64 + lane;
    const f4_t bi = row < m ? ob[row] : zero;
    const f4_t bc = col < m ? ob[col] : zero;                       // lane j holds column box j (a zero box overlaps nothing)
    const float ai = box_area(bi), ac = box_area(bc);
    unsigned long long w = 0ull;
    for (int j = 0; j < 64; ++j) {
      const f4_t bj = f4_t{rl_f(bc[0], j), rl_f(bc[1], j), rl_f(bc[2], j), rl_f(bc[3], j)};
      if (iou_gt(bi, ai, bj, rl_f(ac, j), thr)) w |= 1ull << j;
    }
    if (cc == rc) w &= lane < 63 ? ~0ull << (lane + 1) : 0ull;      // the diagonal tile: strictly later boxes
    if (row < m) M[(int64_t)row * MS_RS + cc] = w;
  }
}

// OR over the 64 lanes, result wave-uniform (scalar): four DPP row rotations leave every lane with the OR of its 16-lane row, four
// readlanes + scalar ORs combine the rows (a __shfl_xor butterfly is 12 dependent LDS-crossbar permutes: ~0.25 us per chunk)
__device__ __forceinline__ unsigned int row_or32(unsigned int v) {
  v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false);     // row_ror:1
  v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false);     // row_ror:2
  v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);     // row_ror:4
  v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);     // row_ror:8
  return v;
}
__device__ __forceinline__ unsigned long long wave_or64(unsigned long long v) {
  const unsigned int lo = row_or32((unsigned int)v), hi = row_or32((unsigned int)(v >> 32));
  unsigned int slo = 0, shi = 0;
#pragma unroll
  for (int r = 0; r < 64; r += 16) {
    slo |= (unsigned int)__builtin_amdgcn_readlane((int)lo, r);
    shi |= (unsigned int)__builtin_amdgcn_readlane((int)hi, r);
  }
  return ((unsigned long long)shi << 32) | slo;
}
typedef __attribute__((address_space(1))) unsigned long long g_u64_t;
__device__ __forceinline__ unsigned long long ldg_u64(const unsigned long long* p) { return *(const g_u64_t*)p; }

__global__ __launch_bounds__(64) void nms_mscan_kernel(int* aux, const unsigned long long* mask, const unsigned long long* skey, int max_det,
                                                       int* kpos_out, unsigned long long* kkey_out) {
  __shared__ int kpos[MS_KR * 64];
  const int b = blockIdx.x, sg = blockIdx.y, lane = threadIdx.x;
  int* ax = aux + (int64_t)b * NMS_AUX_INTS;
  if (!ax[NMS_AUX_OK] || sg >= ax[NMS_AUX_NSEG]) return;
  const int s0 = ax[NMS_AUX_SEG + 2 * sg];
  const int m = s0 + ax[NMS_AUX_SEG + 2 * sg + 1];                   // one past the segment's last candidate slot
  const unsigned long long* M = mask + (int64_t)b * SORT_MAX * MS_RS;
  const int w0 = s0 >> 6, words = (m + 63) >> 6;                     // the segment's chunks: [w0, words)
  unsigned long long G[MS_P][MS_KR], B[MS_P][MS_P + 1], near[MS_P];
  int total = 0;
  auto issue = [&](int c, unsigned long long* g, unsigned long long* bd, int tot) {
    int row = c * 64 + lane;
    if (row > SORT_MAX - 1) row = SORT_MAX - 1;                     // (past the list: any in-bounds row, the words are never used)
    const unsigned long long* rp = M + (int64_t)row * MS_RS + c;
#pragma unroll
    for (int d = 0; d <= MS_P; ++d) bd[d] = ldg_u64(rp + d);
#pragma unroll
    for (int r = 0; r < MS_KR; ++r) {
      const int t = lane + 64 * r;
      const bool ok = t < tot;
      const int k = kpos[ok ? t : 0];
      g[r] = ldg_u64(ok ? M + (int64_t)k * MS_RS + c : reinterpret_cast<const unsigned long long*>(zero_page()));
    }
  };
#pragma unroll
  for (int s = 0; s < MS_P; ++s) { issue(w0 + s, G[s], B[s], 0); near[s] = 0ull; }
  bool done = false;
  for (int c0 = w0; c0 < words && !done; c0 += MS_P) {
#pragma unroll
    for (int s = 0; s < MS_P; ++s) {
      const int c = c0 + s;
      if (c >= words || done) continue;                             // uniform
      const int lim = m - c * 64 < 64 ? m - c * 64 : 64;
      const unsigned long long limmask = lim == 64 ? ~0ull : ((1ull << lim) - 1ull);
      unsigned long long g = G[s][0];
#pragma unroll
      for (int r = 1; r < MS_KR; ++r) g |= G[s][r];
      unsigned long long rem = wave_or64(g) | near[s] | ~limmask;
      unsigned long long nacc[MS_P];
#pragma unroll
      for (int d = 0; d < MS_P; ++d) nacc[d] = 0ull;
      unsigned long long keepbits = 0ull;
      int budget = max_det - total;
      while (~rem != 0ull && budget > 0) {
        const int i = __ffsll((long long)~rem) - 1;
        keepbits |= 1ull << i;
        --budget;
        rem |= rl_u64(B[s][0], i) | (1ull << i);
#pragma unroll
        for (int d = 0; d < MS_P; ++d) nacc[d] |= rl_u64(B[s][d + 1], i);            // this survivor's words c+1 .. c+MS_P
      }
      // word c+d joins accumulator (s+d) % MS_P; d = MS_P re-uses this chunk's own (consumed) accumulator
#pragma unroll
      for (int d = 1; d < MS_P; ++d) near[(s + d) % MS_P] |= nacc[d - 1];
      near[s] = nacc[MS_P - 1];
      if ((keepbits >> lane) & 1ull) kpos[total + __popcll(keepbits & ((1ull << lane) - 1ull))] = c * 64 + lane;
      total += __popcll(keepbits);
      if (total >= max_det) { done = true; continue; }
      issue(c + MS_P, G[s], B[s], total);
    }
  }
  // this segment's kept slots (in keep order = descending score inside the class) join the image's kept list
  int base = 0;
  if (lane == 0 && total > 0) base = atomicAdd(ax + NMS_AUX_KCOUNT, total);
  base = __builtin_amdgcn_readfirstlane(base);
  const unsigned long long* sk = skey + (int64_t)b * SORT_MAX;
  for (int e = lane; e < total; e += 64) {
    const int k = kpos[e];
    kpos_out[(int64_t)b * SORT_MAX + base + e] = k;
    kkey_out[(int64_t)b * SORT_MAX + base + e] = sk[k];
  }
}

// the kept boxes of all segments in descending (score, lower original row) order; the first max_det are the result (general.py:494-495)
__global__ __launch_bounds__(256) void nms_merge_kernel(const int* aux, const int* kpos, const unsigned long long* kkey, const float* sorted,
                                                        int max_nms, int max_det, float* out, int* nkeep) {
  __shared__ unsigned long long sk[256];
  const int b = blockIdx.y;
  const int* ax = aux + (int64_t)b * NMS_AUX_INTS;
  if (!ax[NMS_AUX_OK]) return;
  const int K = ax[NMS_AUX_KCOUNT];
  if (blockIdx.x == 0 && threadIdx.x == 0) nkeep[b] = K < max_det ? K : max_det;
  if ((int)blockIdx.x * 256 >= K) return;
  const int tid = threadIdx.x, i = blockIdx.x * 256 + tid;
  const unsigned long long* kb = kkey + (int64_t)b * SORT_MAX;
  const unsigned long long ki = i < K ? kb[i] : ~0ull;
  int cnt = 0;
  for (int jt = 0; jt < K; jt += 256) {
    sk[tid] = jt + tid < K ? kb[jt + tid] : 0ull;
    __syncthreads();
    const int lim = K - jt < 256 ? K - jt : 256;
    for (int q = 0; q < lim; ++q) cnt += sk[q] > ki ? 1 : 0;
    __syncthreads();
  }
  if (i < K && cnt < max_det) {
    const float* src = sorted + ((int64_t)b * max_nms + kpos[(int64_t)b * SORT_MAX + i]) * 6;
    float* dst = out + ((int64_t)b * max_det + cnt) * 6;
#pragma unroll
    for (int e = 0; e < 6; ++e) dst[e] = src[e];
  }
}

// the launch sequence's counters back to zero in ONE launch (two hipMemsetAsync commands cost the detect.py frame ~30 us of queue bubbles between
// Detect's decode and the filter kernel, profiles/r6_infer_fork_*.txt: rocclr fill commands do not pack behind kernels)
__global__ __launch_bounds__(256) void nms_clear_kernel(int* __restrict__ counts, int batch, int* __restrict__ aux, int naux) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < batch + naux; i += gridDim.x * 256) {
    if (i < batch) counts[i] = 0;
    else aux[i - batch] = 0;
  }
}

}  // namespace

int g_nms_dbg = 0;      // myolo_set_option("nms_dbg", bits): profiling only

extern "C" int64_t myolo_nms_ws_bytes(int batch, int cap) {
  if (batch < 1 || cap < 1) return 0;
  // class-offset boxes | bit matrix | sort keys | score keys by slot | kept keys | ranks | kept slots | per-image side data (NmsAux)
  return (int64_t)batch * ((int64_t)SORT_MAX * 16 + (int64_t)SORT_MAX * MS_RS * 8 + (int64_t)cap * 8 + (int64_t)SORT_MAX * 16 + (int64_t)cap * 4 +
                           (int64_t)SORT_MAX * 4 + (int64_t)NMS_AUX_INTS * 4);
}

extern "C" int myolo_nms(const void* pred, int dtype, int batch, int A, int no, float conf_thres, float iou_thres,
                         int multi_label, int agnostic, float max_wh, int max_nms, int max_det, int cap, int32_t* counts,
                         float* cand, int32_t* cand_idx, float* sorted, float* out, int32_t* nkeep, uint64_t class_mask,
                         int32_t* sort_ws, void* mws, int64_t mws_bytes, void* stream) {
  if (class_mask && no - 5 > 64) return MYOLO_EINVAL;
  if (!pred || (dtype != MYOLO_F16 && dtype != MYOLO_F32) || batch < 1 || A < 1 || no < 6 || cap < 1 || max_det < 1 ||
      max_det > KEPT_MAX || max_nms < 1 || max_nms > MAX_SORTED || !counts || !cand || !cand_idx || !sorted || !out || !nkeep)
    return MYOLO_EINVAL;
  static_assert(RANK_SKIP_BELOW == SORT_MAX && SORT_MAX == (1 << SLOT_BITS), "LDS sort key layout");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipSuccess;
  // lists of up to SORT_MAX candidates (single label, cap == A): rank / scatter / bit matrix on the whole device + the one-wave scan
  const bool matrix = !sort_ws && mws && !((uintptr_t)mws & 15) && mws_bytes >= myolo_nms_ws_bytes(batch, cap) && cap == A &&
                      (int64_t)cap <= (1ll << IDX_BITS) && max_det <= 64 * MS_KR && !(g_nms_dbg & 16);
  f4_t* obx = reinterpret_cast<f4_t*>(mws);
  unsigned long long* mask = matrix ? reinterpret_cast<unsigned long long*>(obx + (int64_t)batch * SORT_MAX) : nullptr;
  unsigned long long* keys = matrix ? mask + (int64_t)batch * SORT_MAX * MS_RS : nullptr;
  unsigned long long* skey = matrix ? keys + (int64_t)batch * cap : nullptr;
  unsigned long long* kkey = matrix ? skey + (int64_t)batch * SORT_MAX : nullptr;
  int* rank = matrix ? reinterpret_cast<int*>(kkey + (int64_t)batch * SORT_MAX) : nullptr;
  int* kpos = matrix ? rank + (int64_t)batch * cap : nullptr;
  int* aux = matrix ? kpos + (int64_t)batch * SORT_MAX : nullptr;
  // class on top of the sort key (classes are then suppressed side by side): single label (cap == A), class id and row must fit the key
  const int keymode = (matrix && !agnostic && no - 5 <= NMS_MAXC && A < (1 << 24) && !(g_nms_dbg & 32)) ? 1 : 0;
  {
    const int naux = matrix ? batch * NMS_AUX_INTS : 0;
    hipLaunchKernelGGL(nms_clear_kernel, dim3(grid_for(batch + naux, 256, 64)), dim3(256), 0, st, counts, batch, aux, naux);
  }
  const int es = dtype == MYOLO_F16 ? 2 : 4;
  const int filter_smem = 256 * no * es + 32;
  if (((uintptr_t)pred & 15) || filter_smem > 150 * 1024) return MYOLO_EINVAL;      // 256 rows of the prediction in LDS (80 classes fp32: 87 KB)
  MYOLO_ENSURE_DYN_SMEM(nms_filter_kernel, filter_smem);
  hipLaunchKernelGGL(nms_filter_kernel, dim3(grid_for(A, 256, 1024), batch), dim3(256), filter_smem, st, pred, dtype, A, no, conf_thres,
                     multi_label, cap, counts, cand, cand_idx, class_mask, keys, rank, (int64_t)batch * A * no * es, aux, keymode);
  if (matrix) {
    hipLaunchKernelGGL(nms_rank2_kernel, dim3(SORT_MAX / 256, 16, batch), dim3(256), 0, st, counts, keys, cap, rank, aux, keymode, max_wh);
    hipLaunchKernelGGL(nms_scatter_kernel, dim3(SORT_MAX / 256, batch), dim3(256), 0, st, counts, cand, rank, keys, cap, max_nms, max_wh,
                       agnostic, keymode, aux, sorted, obx, skey);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(1024, batch), dim3(256), 0, st, aux, obx, mask, iou_thres);
    hipLaunchKernelGGL(nms_mscan_kernel, dim3(batch, keymode ? NMS_MAXC : 1), dim3(64), 0, st, aux, mask, skey, max_det, kpos, kkey);
    hipLaunchKernelGGL(nms_merge_kernel, dim3(SORT_MAX / 256, batch), dim3(256), 0, st, aux, kpos, kkey, sorted, max_nms, max_det, out, nkeep);
  }
  // longer lists (and callers without the matrix workspace): the lazy scan; it orders lists of up to SORT_MAX candidates itself
  // (original rows must fit the key's 19 bits)
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
                       lds_sort | (matrix ? 1 : 0));
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
                     max_det, iou_thres, max_wh, agnostic, lds_sort, out, nkeep, g_nms_dbg, matrix ? aux : (const int*)nullptr);
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

