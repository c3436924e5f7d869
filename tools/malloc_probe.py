import time, torch
torch.cuda.init()
x = torch.empty(1, device="cuda"); torch.cuda.synchronize()
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
for gb in (0.25, 1, 4, 8, 16):
    n = int(gb * (1 << 30))
    p = ctypes.c_void_p()
    t = time.perf_counter(); rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n)); hip.hipDeviceSynchronize(); t1 = time.perf_counter()
    rc2 = hip.hipMemset(p, 0, ctypes.c_size_t(n)); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
    rc3 = hip.hipFree(p); hip.hipDeviceSynchronize(); t3 = time.perf_counter()
    print("%.2f GiB: hipMalloc %.1f ms, first memset %.1f ms, hipFree %.1f ms (rc %d %d %d)" % (gb, (t1-t)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, rc, rc2, rc3), flush=True)
