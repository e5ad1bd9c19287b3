"""Runs tools/ubench/store_inflight.hip on the GPU box (perf exploration; see the .hip header)."""
import ctypes as C
import os

import torch

here = os.path.dirname(os.path.abspath(__file__))
L = C.CDLL(os.path.join(here, "libstore_inflight.so"))
L.run.restype = C.c_float
L.run.argtypes = [C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_void_p), C.c_int, C.c_int]
total = 107 * 1024 * 1024 // 8
bufs = [torch.empty(total + 64, dtype=torch.float64, device="cuda") for _ in range(5)]
ptrs = (C.c_void_p * 5)(*[b.data_ptr() for b in bufs])
torch.cuda.synchronize()
for wpc in (1, 2, 4, 8, 16, 32):
    for width in (8, 16):
        t = L.run(width, wpc * 256, total, ptrs, 5, 20)
        print("%2d wave(s)/CU, %2d B/lane: %8.2f us  %7.1f GB/s" % (wpc, width, t, total * 8 / t / 1e3))
