import ctypes as C, threading, time, numpy as np
hip = C.CDLL("libamdhip64.so")
def chk(r):
    assert r == 0, r
n = 1 << 27  # 1 GiB of doubles
hosts = [np.ones(n) for _ in range(2)]
devs = []
for i in range(2):
    p = C.c_void_p(); chk(hip.hipMalloc(C.byref(p), C.c_size_t(n * 8))); devs.append(p)
streams = []
for i in range(2):
    s = C.c_void_p(); chk(hip.hipStreamCreateWithFlags(C.byref(s), 1)); streams.append(s)
H2D, D2H = 1, 2
def copy(i, kind, parts=1):
    sz = n * 8 // parts
    for j in range(parts):
        if kind == H2D:
            chk(hip.hipMemcpyAsync(C.c_void_p(devs[i].value + j * sz), C.c_void_p(hosts[i].ctypes.data + j * sz), C.c_size_t(sz), kind, streams[i]))
        else:
            chk(hip.hipMemcpyAsync(C.c_void_p(hosts[i].ctypes.data + j * sz), C.c_void_p(devs[i].value + j * sz), C.c_size_t(sz), kind, streams[i]))
    chk(hip.hipStreamSynchronize(streams[i]))
def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
    return best
gb = n * 8 / 1e9
for kind, nm in ((H2D, "H2D"), (D2H, "D2H")):
    for parts in (1, 16, 128):
        t = timed(lambda: copy(0, kind, parts)); print("%s 1 thread  %3d pieces: %.1f GB/s" % (nm, parts, gb / t), flush=True)
    def two():
        th = [threading.Thread(target=copy, args=(i, kind, 16)) for i in range(2)]
        [x.start() for x in th]; [x.join() for x in th]
    t = timed(two); print("%s 2 threads  16 pieces each: %.1f GB/s aggregate" % (nm, 2 * gb / t), flush=True)
def duplex():
    th = [threading.Thread(target=copy, args=(0, H2D, 16)), threading.Thread(target=copy, args=(1, D2H, 16))]
    [x.start() for x in th]; [x.join() for x in th]
t = timed(duplex); print("duplex H2D + D2H: %.1f GB/s aggregate" % (2 * gb / t), flush=True)
# registered (page-locked) host memory
for i in range(2):
    t0 = time.perf_counter(); chk(hip.hipHostRegister(C.c_void_p(hosts[i].ctypes.data), C.c_size_t(n * 8), 0)); print("hipHostRegister 1 GiB: %.1f ms" % (1e3 * (time.perf_counter() - t0)))
for kind, nm in ((H2D, "H2D"), (D2H, "D2H")):
    t = timed(lambda: copy(0, kind, 16)); print("%s registered: %.1f GB/s" % (nm, gb / t), flush=True)
t = timed(duplex); print("duplex registered: %.1f GB/s aggregate" % (2 * gb / t), flush=True)
