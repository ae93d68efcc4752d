"""the detect.py frame's launch pattern without the model: graph A on the main stream, then graph C on a side stream behind an event and graph B
on the main stream, a host sync per iteration.  Ideal time = A + max(B, C).  Which part of the pattern costs the ~150 us the frame's main queue
idles behind graph A (profiles/r6_infer_timeline_*)?  usage: python scripts/ubench/two_queue_gap.py"""
import time
import torch

dev = torch.device('cuda:0')
x = [torch.ones(8 << 20, device=dev, dtype=torch.float16) for _ in range(3)]          # 16 MB each: ~10 us per pass
main = torch.cuda.current_stream()
side = torch.cuda.Stream()


def body(t, n):
    for _ in range(n):
        t.mul_(1.0)


def graph(t, n, stream=None):
    g = torch.cuda.CUDAGraph()
    if stream is None:
        with torch.cuda.graph(g):
            body(t, n)
    else:
        with torch.cuda.graph(g, stream=stream):
            body(t, n)
    return g


NA, NB, NC = 40, 10, 20
for t in x:
    body(t, 3)
with torch.cuda.stream(side):
    body(x[2], 3)
torch.cuda.synchronize()
gA, gB, gC = graph(x[0], NA), graph(x[1], NB), graph(x[2], NC, side)
ev, done = torch.cuda.Event(), torch.cuda.Event()
pre = torch.cuda.Event()


def run_side(g_or_n):
    with torch.cuda.stream(side):
        if isinstance(g_or_n, int):
            body(x[2], g_or_n)
        else:
            g_or_n.replay()
        done.record(side)


def p_main_only():
    gA.replay(); gB.replay()


def p_cb():
    gA.replay(); ev.record(main); side.wait_event(ev); run_side(gC); gB.replay(); main.wait_event(done)


def p_bc():
    gA.replay(); ev.record(main); gB.replay(); side.wait_event(ev); run_side(gC); main.wait_event(done)


def p_cb_side_eager():
    gA.replay(); ev.record(main); side.wait_event(ev); run_side(NC); gB.replay(); main.wait_event(done)


def p_cb_tail_eager():
    gA.replay(); ev.record(main); side.wait_event(ev); run_side(gC); body(x[1], NB); main.wait_event(done)


def p_all_eager():
    body(x[0], NA); ev.record(main); side.wait_event(ev); run_side(NC); body(x[1], NB); main.wait_event(done)


def p_cb_nodep():
    pre.record(main); gA.replay(); side.wait_event(pre); run_side(gC); gB.replay(); main.wait_event(done)


def p_cb_nojoin():
    gA.replay(); ev.record(main); side.wait_event(ev); run_side(gC); gB.replay()


import ctypes as C
hip = C.CDLL('libamdhip64.so')
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipStreamWriteValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint]
hip.hipStreamWaitValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint, C.c_uint32]
sig = C.c_void_p()
import sys
e = hip.hipExtMallocWithFlags(C.byref(sig), 8, 0x2)                       # hipMallocSignalMemory (8 bytes exactly)
print('hipExtMallocWithFlags(signal memory) ->', e)
if e != 0 or 'plain' in sys.argv:
    sig_t = torch.zeros(16, dtype=torch.int32, device=dev)
    sig = C.c_void_p(sig_t.data_ptr())
    print('using plain device memory for the value')
else:
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    assert hip.hipMemset(sig, 0, 8) == 0
torch.cuda.synchronize()
cnt = [0]


def p_memop_bc():
    cnt[0] += 1
    gA.replay()
    assert hip.hipStreamWriteValue32(C.c_void_p(main.cuda_stream), sig, cnt[0], 0) == 0
    gB.replay()
    assert hip.hipStreamWaitValue32(C.c_void_p(side.cuda_stream), sig, cnt[0], 0, 0xffffffff) == 0     # >=
    run_side(gC)
    main.wait_event(done)


def p_memop_cb():
    cnt[0] += 1
    gA.replay()
    assert hip.hipStreamWriteValue32(C.c_void_p(main.cuda_stream), sig, cnt[0], 0) == 0
    assert hip.hipStreamWaitValue32(C.c_void_p(side.cuda_stream), sig, cnt[0], 0, 0xffffffff) == 0
    run_side(gC)
    gB.replay()
    main.wait_event(done)


def p_memop_nojoin():
    cnt[0] += 1
    gA.replay()
    assert hip.hipStreamWriteValue32(C.c_void_p(main.cuda_stream), sig, cnt[0], 0) == 0
    gB.replay()
    assert hip.hipStreamWaitValue32(C.c_void_p(side.cuda_stream), sig, cnt[0], 0, 0xffffffff) == 0
    run_side(gC)


def timeit(f, n=200):
    for _ in range(20):
        f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def solo(g):
    return timeit(lambda: g.replay())


a, b = solo(gA), solo(gB)
with torch.cuda.stream(side):
    pass
c = timeit(lambda: run_side(gC))
print(f'graphs alone (replay + sync): A {a:.1f} us, B {b:.1f} us, C (side) {c:.1f} us; ideal A + max(B, C) ~ {a + max(b, c) - min(a, b, c) * 0:.1f} us minus one sync')
for name, f in (('A, B on the main stream only', p_main_only), ('A | C side, B main (order c, b)', p_cb), ('A | B main, C side (order b, c)', p_bc),
                ('order c, b, C as eager launches', p_cb_side_eager), ('order c, b, B as eager launches', p_cb_tail_eager), ('everything eager', p_all_eager),
                ('order c, b, side waits for an event recorded BEFORE A (no dependency on A)', p_cb_nodep), ('order c, b, main does not join C', p_cb_nojoin),
                ('hipStreamWriteValue32 behind A / hipStreamWaitValue32 in front of C, order b, c', p_memop_bc), ('the same, order c, b', p_memop_cb),
                ('the same, order b, c, main does not join C', p_memop_nojoin)):
    print(f'{timeit(f):8.1f} us  {name}')
