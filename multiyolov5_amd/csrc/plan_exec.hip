// Native executor of a static launch plan: ONE C call walks a serialised launch list (the forward or the backward of a
// multiyolov5_amd.engine.Plan: ~330 + ~340 libmyolo launches for yolov5s+PSP) instead of one ctypes call per launch from Python
// (round 2: 8.0 ms of host time per 9.3 ms step, DESIGN section 7 lead 0).
//
//   * an op is a fixed-size record: kind, function id, 8-byte argument slots (pointers, integers sign-extended, floats as their bit
//     pattern in the low word).  The typed call is rebuilt by a per-entry-point thunk generated from the function's own signature
//     (template over the argument pack), so the records carry no type tags and a new entry point costs one REG() line;
//   * descriptors (myolo_conv_desc ...) are built once by the host mirror and referenced by address; the few slots that change per
//     run (the caller's input tensors) are patched in place through myolo_prog_slot();
//   * weight gradients fork to the side stream exactly like the Python loop did: event record on the main stream, wait on the side
//     stream, launch there; a JOIN makes the main stream wait for the side stream.  Events are created once per program;
//   * a run is a [first, last) range of ops, so the caller can interleave its RCCL gradient slices (parallel.GradReducer) between
//     ranges.  Nothing here allocates, synchronises or touches the host after create: hipGraph-capturable like the launches themselves.
//
// Not a reference interface: the reference's forward is Python (models/yolo.py:293-316); this replaces the interpreter loop around it.
#include "myolo_dev.h"
#include <stdlib.h>
#include <string.h>
#include <tuple>
#include <utility>
#include <vector>

namespace {

template <class T> struct Unpack {
  static T get(uint64_t v) {
    if constexpr (std::is_pointer<T>::value) return reinterpret_cast<T>(static_cast<uintptr_t>(v));
    else if constexpr (std::is_same<T, float>::value) { const uint32_t b = (uint32_t)v; float f; memcpy(&f, &b, 4); return f; }
    else return static_cast<T>(static_cast<int64_t>(v));
  }
};

template <class... Args, size_t... I>
int call_impl(int (*fn)(Args...), const uint64_t* a, void* st, std::index_sequence<I...>) {
  using Tup = std::tuple<Args...>;
  return fn(Unpack<std::tuple_element_t<I, Tup>>::get(a[I])..., st);
}
template <class... Args>
int call_packed(int (*fn)(Args...), const uint64_t* a, void* st) {
  static_assert(sizeof...(Args) >= 1 && sizeof...(Args) - 1 <= MYOLO_PROG_MAX_ARGS, "argument slots");
  return call_impl(fn, a, st, std::make_index_sequence<sizeof...(Args) - 1>{});
}

typedef int (*thunk_t)(const uint64_t*, void*);
struct Entry { const char* name; thunk_t fn; int nargs; };
template <class... Args> constexpr int nargs_of(int (*)(Args...)) { return (int)sizeof...(Args) - 1; }

#define REG(f) {#f, [](const uint64_t* a, void* st) -> int { return call_packed(&f, a, st); }, nargs_of(&f)}
const Entry g_table[] = {
    REG(myolo_pack_weight), REG(myolo_pack_weights_mt), REG(myolo_focus_pack), REG(myolo_conv), REG(myolo_conv_dgrad_s2), REG(myolo_conv_dgrad_bn), REG(myolo_conv_pair), REG(myolo_conv_bn_act),
    REG(myolo_conv_wgrad), REG(myolo_bn_wgrad_stem), REG(myolo_bn_act_fwd), REG(myolo_bn_act_bwd_reduce), REG(myolo_bn_act_bwd_apply),
    REG(myolo_bn_act_fwd_split), REG(myolo_bn_act_bwd_reduce_split), REG(myolo_bn_act_bwd_apply_split), REG(myolo_bn_act_bwd_fused), REG(myolo_spp_pool_fwd),
    REG(myolo_spp_pool_bwd), REG(myolo_copy_up_fwd), REG(myolo_copy_up_bwd), REG(myolo_bilinear_fwd), REG(myolo_bilinear_bwd),
    REG(myolo_adaptive_avgpool_fwd), REG(myolo_adaptive_avgpool_fwd_multi), REG(myolo_adaptive_avgpool_bwd), REG(myolo_adaptive_avgpool_bwd_multi),
    REG(myolo_pyramid_upsample_fwd), REG(myolo_pyramid_upsample_bwd), REG(myolo_gate_fwd), REG(myolo_gate_bwd),
    REG(myolo_gate_mul_fwd), REG(myolo_gate_mul_bwd), REG(myolo_add), REG(myolo_fill_zero), REG(myolo_cast_from_f32),
    REG(myolo_dropout_fwd), REG(myolo_dropout_bwd), REG(myolo_seg_upsample_fwd), REG(myolo_seg_upsample_bwd),
    REG(myolo_seg_lowgrad_apply), REG(myolo_detect_unpermute), REG(myolo_detect_decode), REG(myolo_tiny_conv_fwd), REG(myolo_tiny_conv_bwd),
};
#undef REG
constexpr int NFN = (int)(sizeof(g_table) / sizeof(g_table[0]));

struct Prog {
  std::vector<myolo_prog_op> ops;
  std::vector<hipEvent_t> evs;        // one per op that needs one (FORK / JOIN), else nullptr
  int last_op = -1;
};

}  // namespace

extern "C" int myolo_prog_fn_id(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < NFN; ++i)
    if (!strcmp(g_table[i].name, name)) return i;
  return -1;
}

extern "C" int myolo_prog_fn_nargs(int fn) { return (fn >= 0 && fn < NFN) ? g_table[fn].nargs : -1; }

extern "C" void* myolo_prog_create(const myolo_prog_op* ops, int n) {
  if (!ops || n < 0) return nullptr;
  for (int i = 0; i < n; ++i) {
    const myolo_prog_op& o = ops[i];
    if (o.kind < MYOLO_OP_CALL || o.kind > MYOLO_OP_MEMSET) return nullptr;
    if ((o.kind == MYOLO_OP_CALL || o.kind == MYOLO_OP_CALL_SIDE) && (o.fn < 0 || o.fn >= NFN || o.nargs != g_table[o.fn].nargs)) return nullptr;
  }
  Prog* p = new Prog;
  p->ops.assign(ops, ops + n);
  p->evs.assign(n, nullptr);
  for (int i = 0; i < n; ++i)
    if (ops[i].kind == MYOLO_OP_CALL_SIDE || ops[i].kind == MYOLO_OP_JOIN) {
      // no timing, no system-scope fence: the event only orders two streams of this device (a default event's record releases to
      // system scope: 5-8 us of main-queue idle per weight-gradient fork in the r3d trace, 79 forks per step)
      if (hipEventCreateWithFlags(&p->evs[i], hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) {
        for (hipEvent_t e : p->evs) if (e) (void)hipEventDestroy(e);
        delete p;
        return nullptr;
      }
    }
  return p;
}

extern "C" void myolo_prog_destroy(void* prog) {
  Prog* p = static_cast<Prog*>(prog);
  if (!p) return;
  for (hipEvent_t e : p->evs) if (e) (void)hipEventDestroy(e);
  delete p;
}

extern "C" uint64_t* myolo_prog_slot(void* prog, int op, int arg) {
  Prog* p = static_cast<Prog*>(prog);
  if (!p || op < 0 || op >= (int)p->ops.size() || arg < 0 || arg >= MYOLO_PROG_MAX_ARGS) return nullptr;
  return &p->ops[op].a[arg];
}

extern "C" int myolo_prog_last_op(void* prog) { return prog ? static_cast<Prog*>(prog)->last_op : -1; }

// MEMSET ops are a kernel of this library, not hipMemsetAsync: (1) a memset NODE of a captured graph is re-issued through the runtime's blit path
// on every replay, and in the detect.py loop (graph replays + NMS's own launches in between) it stopped clearing the forward's fp32 accumulators after
// a few hundred frames in about half of the processes -- whatever the fork mechanism, until the runtime's kernel-argument pool wrapped the next time
// (scripts/ubench/fork_stress.py, profiles/r6_fork_stress.txt: the segmentation head's pooled sums grew to inf); a kernel node carries its own
// arguments; (2) rocclr fill commands do not pack behind kernels: ~4 us of queue bubble each.
namespace {
struct FillR { uint4* p16; size_t n16; unsigned int* tail; int ntail; unsigned int v; };
__global__ __launch_bounds__(256) void fill_words_kernel(FillR a, FillR b) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
  const uint4 va = {a.v, a.v, a.v, a.v}, vb = {b.v, b.v, b.v, b.v};
  for (size_t i = t; i < a.n16; i += step) a.p16[i] = va;
  for (size_t i = t; i < b.n16; i += step) b.p16[i] = vb;
  if (t < (size_t)a.ntail) a.tail[t] = a.v;
  if (t < (size_t)b.ntail) b.tail[t] = b.v;
}
FillR fill_region(void* ptr, size_t bytes, unsigned int v) {
  FillR r{nullptr, 0, nullptr, 0, v};
  if (!ptr || !bytes) return r;
  char* p = static_cast<char*>(ptr);
  size_t words = bytes >> 2;
  size_t head = ((uintptr_t)p & 15) ? (16 - ((uintptr_t)p & 15)) >> 2 : 0;       // words up to the first 16-byte boundary
  if (head > words) head = words;
  if (head || words < 4) {                       // small or unaligned: everything through the word path (<= 256 words), else split
    if (words <= 256) { r.tail = reinterpret_cast<unsigned int*>(p); r.ntail = (int)words; return r; }
  }
  // aligned body + word tail (callers pass 16-byte aligned bases for anything larger than 1 KB)
  r.p16 = reinterpret_cast<uint4*>(p); r.n16 = words >> 2;
  r.tail = reinterpret_cast<unsigned int*>(p + (r.n16 << 4)); r.ntail = (int)(words & 3);
  return r;
}
}  // namespace

// hipMemsetAsync's replacement inside the library: up to two regions of 32-bit words in ONE launch (value 0 or any word pattern)
int myolo_fill_words2(void* a, size_t a_bytes, unsigned int av, void* b, size_t b_bytes, unsigned int bv, hipStream_t st) {
  if ((a_bytes & 3) || (b_bytes & 3) || ((uintptr_t)a & 3) || ((uintptr_t)b & 3)) return MYOLO_EINVAL;
  // (a large region off the 16-byte grid: the runtime's memset for that one -- byte patterns only; the library's callers never get here)
  if (a_bytes > 1024 && ((uintptr_t)a & 15)) {
    if (av != 0u && av != 0xffffffffu) return MYOLO_EINVAL;
    const hipError_t e = hipMemsetAsync(a, (int)(av & 0xff), a_bytes, st);
    if (e != hipSuccess) return (int)e;
    a = nullptr; a_bytes = 0;
  }
  if (b_bytes > 1024 && ((uintptr_t)b & 15)) {
    if (bv != 0u && bv != 0xffffffffu) return MYOLO_EINVAL;
    const hipError_t e = hipMemsetAsync(b, (int)(bv & 0xff), b_bytes, st);
    if (e != hipSuccess) return (int)e;
    b = nullptr; b_bytes = 0;
  }
  const FillR ra = fill_region(a, a_bytes, av), rb = fill_region(b, b_bytes, bv);
  const size_t work = ra.n16 > rb.n16 ? ra.n16 : rb.n16;
  if (!work && !ra.ntail && !rb.ntail) return 0;
  hipLaunchKernelGGL(fill_words_kernel, dim3(grid_for((int64_t)(work ? work : 1), 256, 2048)), dim3(256), 0, st, ra, rb);
  return (int)hipGetLastError();
}
static int zero_fill(void* ptr, size_t bytes, hipStream_t st) {
  if (!bytes) return 0;
  if (!ptr || ((uintptr_t)ptr & 15) || (bytes & 3)) return (int)hipMemsetAsync(ptr, 0, bytes, st);     // (never the plans' arenas: 256-byte aligned fp32)
  return myolo_fill_words2(ptr, bytes, 0u, nullptr, 0, 0u, st);
}

// (round 5, measured and removed: forking the weight-gradient stream once per K side calls instead of per call -- MYOLO_SIDE_BATCH=2 / 4 / 8:
//  7.769 / 7.855 / 7.911 ms per step against 7.765; the held-back weight gradients start later and the exposed tail grows)
extern "C" int myolo_prog_run(void* prog, int first, int last, void* main_stream, void* side_stream) {
  Prog* p = static_cast<Prog*>(prog);
  if (!p || first < 0 || last > (int)p->ops.size() || first > last) return MYOLO_EINVAL;
  hipStream_t ms = (hipStream_t)main_stream, ss = (hipStream_t)side_stream;
  for (int i = first; i < last; ++i) {
    const myolo_prog_op& o = p->ops[i];
    if (o.cond && *reinterpret_cast<const int32_t*>(static_cast<uintptr_t>(o.cond)) != o.cond_val) continue;
    int r = 0;
    switch (o.kind) {
      case MYOLO_OP_CALL:
        r = g_table[o.fn].fn(o.a, main_stream);
        break;
      case MYOLO_OP_CALL_SIDE:
        if (ss) {                                   // fork: the side stream picks up behind everything enqueued on the main stream so far
          hipError_t e = hipEventRecord(p->evs[i], ms);
          if (e == hipSuccess) e = hipStreamWaitEvent(ss, p->evs[i], 0);
          r = e != hipSuccess ? (int)e : g_table[o.fn].fn(o.a, side_stream);
        } else {
          r = g_table[o.fn].fn(o.a, main_stream);
        }
        break;
      case MYOLO_OP_JOIN:
        if (ss) {
          hipError_t e = hipEventRecord(p->evs[i], ss);
          if (e == hipSuccess) e = hipStreamWaitEvent(ms, p->evs[i], 0);
          r = (int)e;
        }
        break;
      case MYOLO_OP_MEMSET:
        r = zero_fill(reinterpret_cast<void*>(static_cast<uintptr_t>(o.a[0])), (size_t)o.a[1], ms);
        break;
      default:
        r = MYOLO_EINVAL;
    }
    if (r) { p->last_op = i; return r; }
  }
  return 0;
}

// ---- order between two queues without a HIP event (include/myolo.h: myolo_queue_post / myolo_queue_wait) -------------------------------
// hipEventRecord + hipStreamWaitEvent between two queues costs 90-170 us on this runtime whatever is waited for (scripts/ubench/two_queue_gap.py:
// graph A | graph C on a second stream + graph B behind A: 423 us against ~330 for A + max(B, C); no HIP / ROCr switch moves it, and HIP's own
// stream memory operations -- hipStreamWriteValue32 / hipStreamWaitValue32 -- sit in between at 373-392 us and made the real frame SLOWER, 866
// against 1014 FPS: profiles/r6_two_queue_gap.txt, r6_infer_fork_ab.txt): the detect.py frame's main queue idled that long behind the backbone
// graph.  A counting semaphore in device memory does the same job in about a microsecond: the producer queue runs a one-lane kernel that adds 1
// behind its work, the consumer queue starts with a one-lane kernel that polls (agent-scope loads, s_sleep) until the count is positive and takes
// 1.  What the producer wrote is visible to the consumer's NEXT kernel: every dispatch ends with an agent-scope release and starts with an
// agent-scope acquire (checked the hard way: scripts/ubench/fork_stress.py, 1200 frames whose tensors all stay in the L2s, every frame against
// the one-stream forward).  The poll is bounded (timeout_ms of the 100 MHz clock): on expiry the kernel counts a timeout in sem[32] and lets its
// queue go on -- the host side (runtime.PlanHolder) reads that word and refuses the results.
namespace {
__global__ __launch_bounds__(64) void queue_post_kernel(unsigned int* sem) {
  if (threadIdx.x == 0) __hip_atomic_fetch_add((gbar_u32*)sem, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(64) void queue_wait_kernel(unsigned int* sem, long long limit) {
  if (threadIdx.x != 0) return;
  gbar_u32* w = (gbar_u32*)sem;
  const long long t0 = wall_clock64();
  for (;;) {
    if ((int)__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) {
      __hip_atomic_fetch_sub(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    __builtin_amdgcn_s_sleep(16);
    if (wall_clock64() - t0 > limit) break;
  }
  __hip_atomic_fetch_add(w + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace

extern "C" int myolo_queue_post(void* sem, void* stream) {
  if (!sem || ((uintptr_t)sem & 3)) return MYOLO_EINVAL;
  hipLaunchKernelGGL(queue_post_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, static_cast<unsigned int*>(sem));
  MYOLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int myolo_queue_wait(void* sem, int timeout_ms, void* stream) {
  if (!sem || ((uintptr_t)sem & 3) || timeout_ms < 1) return MYOLO_EINVAL;
  hipLaunchKernelGGL(queue_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, static_cast<unsigned int*>(sem), (long long)timeout_ms * 100000ll);
  MYOLO_CHECK_LAUNCH();
  return 0;
}

// ---- launch trace (test infrastructure; see MYOLO_CHECK_LAUNCH in myolo_dev.h) --------------------------------------------------------
#include <map>
#include <mutex>
#include <string>
int g_myolo_trace = 0;
static std::mutex g_trace_mu;
static std::map<std::string, int> g_trace_sites;      // launch site (kernel family + template arguments) -> launches since myolo_trace_start(1)
void myolo_trace_note(const char* site) {
  std::lock_guard<std::mutex> lk(g_trace_mu);
  ++g_trace_sites[site ? site : "?"];
}
extern "C" int myolo_trace_start(int on) {
  std::lock_guard<std::mutex> lk(g_trace_mu);
  g_trace_sites.clear();
  g_myolo_trace = on ? 1 : 0;
  return 0;
}
extern "C" int64_t myolo_trace_read(char* buf, int64_t cap) {
  std::lock_guard<std::mutex> lk(g_trace_mu);
  std::string out;
  for (const auto& kv : g_trace_sites) { out += std::to_string(kv.second); out += '\t'; out += kv.first; out += '\n'; }
  if (buf && cap > 0) {
    const int64_t n = (int64_t)out.size() < cap - 1 ? (int64_t)out.size() : cap - 1;
    memcpy(buf, out.data(), (size_t)n);
    buf[n] = 0;
  }
  return (int64_t)out.size() + 1;                     // bytes needed (call with buf = NULL to size the buffer)
}
