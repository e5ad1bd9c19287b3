"""Runs tools/ubench/store_pattern.hip on the GPU box (perf exploration; see the .hip header)."""
import ctypes as C
import os
import subprocess
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libstore_pattern.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so,
                           os.path.join(here, "store_pattern.hip")])
L = C.CDLL(so)
L.run.restype = C.c_float
L.run.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_void_p]
N, instances = 4096, 16
nblk = 204   # ~ (852356 / 4096): 16 instances x 204 x 4096 x 8 B = 107 MB per launch
stride = nblk * N + 8
bufs = [torch.empty(instances * stride + 64, dtype=torch.float64, device="cuda") for _ in range(16)]
torch.cuda.synchronize()
for mode, name in [(0, "linear 8B/lane"), (2, "linear 16B/lane"), (1, "pattern 512B runs"), (3, "pattern 1KB runs"), (4, "pattern 512B nontemporal"), (5, "linear nontemporal")]:
    for nb, lds in ((1, 0), (5, 0), (16, 0), (16, 80 * 1024)):
        ptrs = (C.c_void_p * nb)(*[b.data_ptr() for b in bufs[:nb]])
        t = L.run(mode, lds, ptrs, nb, N, nblk, instances, stride, 64, None)
        print("%d buffer(s), %3d KB LDS/WG: " % (nb, lds // 1024), end="")
        print("%-20s %8.2f us  %7.1f GB/s" % (name, t, instances * nblk * N * 8 / t / 1e3))

# alignment of the instance stride: every 512-byte run starts (stride * 8 * inst) bytes into a 128-byte line
print("pattern 512B runs, 16 buffers, instance stride = nblk*N + pad doubles:")
for pad in (0, 16, 8, 4, 1):
    st = nblk * N + pad
    ptrs = (C.c_void_p * 16)(*[b.data_ptr() for b in bufs])
    t = L.run(1, 0, ptrs, 16, N, nblk, instances, st, 64, None)
    print("  pad %2d doubles (%3d B): %8.2f us  %7.1f GB/s" % (pad, (pad * 8) % 128, t, instances * nblk * N * 8 / t / 1e3))

