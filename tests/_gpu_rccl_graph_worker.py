"""Worker of tests/test_z_gpu_exchange.py::test_rccl_all_gather_captured_in_the_step_graph (one process, world size 1)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lpopc_amd import problems  # noqa: E402
from lpopc_amd.dist import IntervalExchange, pack_all_host, slot_layout  # noqa: E402
from lpopc_amd.engine import NLPEngine  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    prob = problems.launch(16, 8)
    B = 4
    eng = NLPEngine(prob, n_instances=B, shard_mode=1, shard_rank=0, shard_world=1, device=0)
    ref = NLPEngine(prob, n_instances=B, device=0)
    xl, xu, _, _ = ref.get_bounds_info()
    x0 = ref.get_starting_point()[:ref.n]
    dx = torch.from_numpy(np.stack([problems.seeded_iterate(x0, xl, xu, 3 + b) for b in range(B)])).cuda()
    mk = lambda n: torch.full((B, n), np.nan, dtype=torch.float64, device="cuda")   # noqa: E731
    dg, dv, rg, rv = mk(eng.m), mk(eng.nnz_jac), mk(eng.m), mk(eng.nnz_jac)
    xch = IntervalExchange(eng, dist, 1, 0)

    def step():
        eng.eval_pair_dev(dx, dg, dv)
        xch.exchange(dg, dv)
    step()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(3):
            step()
    dg.fill_(np.nan)
    dv.fill_(np.nan)
    xch.buf.fill_(np.nan)
    graph.replay()
    ref.eval_pair_dev(dx, rg, rv)
    torch.cuda.synchronize()
    assert torch.equal(dg, rg) and torch.equal(dv, rv)
    # the slot holds exactly this rank's packed runs
    lay, slot = slot_layout(eng, 1, B)
    assert slot == xch.slot
    host = pack_all_host(rg.cpu().numpy().ravel(), rv.cpu().numpy().ravel(), lay[0], B, eng.m, eng.nnz_jac, slot)
    got = xch.buf.cpu().numpy()
    used = B * (lay[0][2] + lay[0][3])
    assert np.array_equal(got[:used], host[:used])
    del graph                      # graphs that hold collective nodes go before the communicator
    torch.cuda.synchronize()
    eng.close()
    ref.close()
    dist.destroy_process_group()
    print("rccl graph ok")


if __name__ == "__main__":
    main()
