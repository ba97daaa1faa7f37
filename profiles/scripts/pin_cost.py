import time, numpy as np, torch
rt = torch.cuda.cudart()
torch.zeros(1, device="cuda")
for rep in range(4):
    a = np.ones(150_000_000, dtype=np.float64)  # 1.2 GB, touched
    t0 = time.perf_counter(); rc = rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0); t1 = time.perf_counter()
    t = torch.empty(150_000_000, dtype=torch.float64, device="cuda")
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    t2 = time.perf_counter(); hip.hipMemcpy(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes), 1); torch.cuda.synchronize(); t3 = time.perf_counter()
    rc2 = rt.cudaHostUnregister(a.ctypes.data); t4 = time.perf_counter()
    print("register %.1f ms (rc %s)  copy %.1f ms  unregister %.1f ms" % ((t1 - t0) * 1e3, rc, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
    t5 = time.perf_counter(); hip.hipMemcpy(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes), 1); torch.cuda.synchronize(); t6 = time.perf_counter()
    print("   pageable copy %.1f ms" % ((t6 - t5) * 1e3))
    del a, t
