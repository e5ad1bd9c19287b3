import ctypes as C, os, numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
L = C.CDLL(os.path.join(here, "liblds_direct.so"))
src = torch.arange(4096, dtype=torch.float64, device="cuda") + 0.5
ok = True
for off, ln in [(0, 128), (1, 128), (3, 81), (5, 1088), (7, 1089), (2, 2), (9, 3), (11, 257)]:
    nout = 9 + ln + 8
    out = torch.zeros(nout, dtype=torch.float64, device="cuda")
    import ctypes
    hip = C.CDLL("libamdhip64.so")
    # launch through hipModule-less path: use the kernel symbol via hipLaunchKernel
    args = (C.c_void_p * 5)()
    a0, a1, a2, a3, a4 = C.c_void_p(src.data_ptr()), C.c_int(off), C.c_int(ln), C.c_void_p(out.data_ptr()), C.c_int(nout)
    for i, a in enumerate((a0, a1, a2, a3, a4)):
        args[i] = C.cast(C.pointer(a), C.c_void_p)
    class dim3(C.Structure):
        _fields_ = [("x", C.c_uint), ("y", C.c_uint), ("z", C.c_uint)]
    f = C.cast(L.k_copy, C.c_void_p)
    rc = hip.hipLaunchKernel(f, dim3(1, 1, 1), dim3(256, 1, 1), args, C.c_size_t(nout * 8), C.c_void_p(0))
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    exp = np.full(nout, -1.0); exp[9:9 + ln] = np.arange(off, off + ln) + 0.5
    good = np.array_equal(o, exp)
    ok &= good
    print(off, ln, rc, "OK" if good else "MISMATCH", "" if good else (np.nonzero(o != exp)[0][:10], o[np.nonzero(o != exp)[0][:10]]))
print("ALL OK" if ok else "FAILED")
