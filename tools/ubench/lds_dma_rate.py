import ctypes as C, os, numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
L = C.CDLL(os.path.join(here, "liblds_dma_rate.so"))
L.run.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
src = torch.arange(64 * 4096, dtype=torch.float64, device="cuda")
blocks = 256
st = torch.zeros(blocks * 4, dtype=torch.int64, device="cuda")
for mode, nm in ((0, "lds-dma x4 (1 KB/instr)"), (1, "lds-dma dword (256 B/instr)"), (2, "regs + ds_write")):
    for nch in (8, 16):
        for waves in (1, 2, 4):
            if mode == 2 and nch > 8 * waves:
                continue
            L.run(mode, src.data_ptr(), st.data_ptr(), blocks, nch, waves)
            t = st.cpu().numpy().reshape(blocks, 4).astype(np.float64) * 0.01
            print("%-28s %2d KB by %d wave(s): issue %.2f us, landed %.2f us" % (nm, nch, waves, (t[:, 1] - t[:, 0]).mean(), (t[:, 2] - t[:, 0]).mean()))
