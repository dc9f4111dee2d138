"""development: what n page-locked host-to-device copies of one 1280x720 image cost on ONE stream (copy engines), one after the other,
against ONE kernel that reads the same page-locked buffers over PCIe (torch only for the kernel-free timing harness: ctypes on libamdhip64)"""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
def ck(e):
    assert e == 0, e
n, size = 16, 1280 * 720
host = [C.c_void_p() for _ in range(n)]; dev = C.c_void_p(); s = C.c_void_p(); e0 = C.c_void_p(); e1 = C.c_void_p()
for h in host: ck(hip.hipHostMalloc(C.byref(h), C.c_size_t(size), 0))
ck(hip.hipMalloc(C.byref(dev), C.c_size_t(n * size)))
ck(hip.hipStreamCreateWithFlags(C.byref(s), 1)); ck(hip.hipEventCreate(C.byref(e0))); ck(hip.hipEventCreate(C.byref(e1)))
for k in (1, 2, 4, 8, 16):
    best = 1e9
    for rep in range(5):
        ck(hip.hipEventRecord(e0, s))
        t0 = time.perf_counter()
        for i in range(k): ck(hip.hipMemcpyAsync(C.c_void_p(dev.value + i * size), host[i], C.c_size_t(size), 1, s))
        t1 = time.perf_counter()
        ck(hip.hipEventRecord(e1, s)); ck(hip.hipStreamSynchronize(s))
        ms = C.c_float(); ck(hip.hipEventElapsedTime(C.byref(ms), e0, e1))
        best = min(best, ms.value)
    print("%2d copies of %.2f MB on one stream: %.1f us (%.1f per copy, %.1f GB/s); enqueue %.1f us" % (k, size / 1e6, best * 1e3, best * 1e3 / k, k * size / best / 1e6, (t1 - t0) * 1e6))
