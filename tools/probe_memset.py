import ctypes as C, time, os, torch
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
torch.zeros(1, device="cuda")
for flags, name in ((0, "cached"), (3, "uncached")):
    p = C.c_void_p()
    n = 12 << 20
    if flags: print("alloc", hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(n), C.c_uint(flags)))
    else: print("alloc", hip.hipMalloc(C.byref(p), C.c_size_t(n)))
    for rep in range(3):
        hip.hipDeviceSynchronize()
        t0 = time.perf_counter()
        hip.hipMemset(p, 0, C.c_size_t(n))
        hip.hipDeviceSynchronize()
        print(name, "hipMemset 12 MB:", round((time.perf_counter() - t0) * 1e3, 2), "ms")
