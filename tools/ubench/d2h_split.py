"""D2H bandwidth of a Jacobian-sized buffer: one copy vs split over several streams (perf exploration)."""
import time
import torch

for mb in (2.75, 6.8):
    n = int(mb * 1e6 / 8)
    dev = torch.randn(n, dtype=torch.float64, device="cuda")
    host = torch.empty(n, dtype=torch.float64).pin_memory()
    for parts in (1, 2, 4, 8):
        streams = [torch.cuda.Stream() for _ in range(parts)]
        chunk = (n + parts - 1) // parts
        def run():
            for p, st in enumerate(streams):
                with torch.cuda.stream(st):
                    host[p * chunk:(p + 1) * chunk].copy_(dev[p * chunk:(p + 1) * chunk], non_blocking=True)
            for st in streams:
                st.synchronize()
        for _ in range(5):
            run()
        t0 = time.perf_counter()
        K = 50
        for _ in range(K):
            run()
        us = (time.perf_counter() - t0) / K * 1e6
        print("%.2f MB in %d part(s): %.1f us  %.1f GB/s" % (mb, parts, us, n * 8 / us / 1e3))
