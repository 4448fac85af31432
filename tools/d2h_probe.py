import ctypes as C, time, numpy as np
hip = C.CDLL("libamdhip64.so")
def chk(r):
    assert r == 0, r
n = 8 << 20
d = C.c_void_p(); chk(hip.hipMalloc(C.byref(d), n)); chk(hip.hipMemset(d, 1, n)); chk(hip.hipDeviceSynchronize())
def t(label, f, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(1e3 * (time.perf_counter() - t0))
    print("%-55s %s" % (label, " ".join("%.2f" % x for x in ts)))
dst = np.empty(n, np.uint8)
t("hipMemcpy D2H -> np.empty (same array, reused)", lambda: chk(hip.hipMemcpy(dst.ctypes.data_as(C.c_void_p), d, n, 2)))
t("hipMemcpy D2H -> fresh np.empty each time", lambda: chk(hip.hipMemcpy(np.empty(n, np.uint8).ctypes.data_as(C.c_void_p), d, n, 2)))
t("hipMemcpy D2H -> fresh np.zeros each time", lambda: chk(hip.hipMemcpy(np.zeros(n, np.uint8).ctypes.data_as(C.c_void_p), d, n, 2)))
for flag, name in ((0, "default"), (0x40000000, "coherent"), (0x80000000, "noncoherent"), (0x1, "portable")):
    h = C.c_void_p(); chk(hip.hipHostMalloc(C.byref(h), n, C.c_uint(flag)))
    t("hipMemcpy D2H -> pinned(%s)" % name, lambda: chk(hip.hipMemcpy(h, d, n, 2)))
    buf = (C.c_uint8 * n).from_address(h.value); arr = np.frombuffer(buf, dtype=np.uint8)
    t("  numpy copy out of pinned(%s)" % name, lambda: arr.copy())
    t("  memmove pinned(%s) -> np.empty reused" % name, lambda: C.memmove(dst.ctypes.data, h.value, n))
    hip.hipHostFree(h)
t("np.empty(8MB).fill(0)", lambda: np.empty(n, np.uint8).fill(0))
s = C.c_void_p(); chk(hip.hipStreamCreateWithFlags(C.byref(s), 1))
t("hipMemsetAsync 32MB + sync on nonblocking stream", lambda: (chk(hip.hipMemsetAsync(d, 0, n, s)), chk(hip.hipStreamSynchronize(s))))
small = C.c_void_p()
def sm():
    chk(hip.hipMalloc(C.byref(small), 1600)); chk(hip.hipFree(small))
t("hipMalloc+hipFree 1.6 KB", sm)
hb = np.zeros(200, np.int64)
t("hipMemcpyAsync H2D 1.6KB pageable + sync", lambda: (chk(hip.hipMemcpyAsync(d, hb.ctypes.data_as(C.c_void_p), 1600, 1, s)), chk(hip.hipStreamSynchronize(s))))
