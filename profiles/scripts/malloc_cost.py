import time, ctypes, torch
torch.zeros(1, device="cuda")
hip = ctypes.CDLL("libamdhip64.so")
def tm(sizes):
    ps = []
    t0 = time.perf_counter()
    for s in sizes:
        p = ctypes.c_void_p()
        rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(s)); assert rc == 0
        ps.append(p)
    t1 = time.perf_counter()
    for p in ps: hip.hipFree(p)
    t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3
G = 1 << 30
for name, sizes in (("10 x 1.2 GB", [int(1.2 * G)] * 10), ("1 x 12 GB", [12 * G]), ("40 x 0.3 GB", [int(0.3 * G)] * 40), ("10 x 1.2 GB again", [int(1.2 * G)] * 10), ("100 x 4 MB", [4 << 20] * 100)):
    a, f = tm(sizes)
    print("%-20s malloc %.2f ms   free %.2f ms" % (name, a, f))
# first kernel touching fresh memory
x = torch.empty(int(1.2 * G) // 8, dtype=torch.float64, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter(); x.zero_(); torch.cuda.synchronize(); t1 = time.perf_counter(); x.zero_(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("first touch of 1.2 GB %.2f ms, second %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
