"""Worker of tests/test_z_gpu_exchange.py::test_two_processes_on_one_gpu_host_consumer_and_sweep: two ranks share GPU 0
(gloo carries the control messages; RCCL refuses two ranks on one device), and rehearse
  * the host-consumer mode: both ranks' interval-sharded engines store their runs of g / values into ONE shared
    page-locked host segment (lpopc_amd.dist.HostConsumerGroup) == a single engine's result, bit for bit, over several
    iterates (so that the delivery by difference is exercised), on a ragged hp mesh and on the 4-phase launch problem;
  * SweepShard around the real device interior-point solver (BatchedIPM): each rank solves its share of a sweep, only the
    verdicts are gathered == one engine solving the whole sweep."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lpopc_amd import problems  # noqa: E402
from lpopc_amd.dist import HostConsumerGroup, SweepShard  # noqa: E402
from lpopc_amd.engine import BatchedIPM, NLPEngine  # noqa: E402
from lpopc_amd.problem import Options  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    for make, mode in ((lambda: problems.launch(16, 8), "perturb"), (lambda: problems.config("hypersensitive"), "uniform")):
        prob = make()
        eng = NLPEngine(prob, shard_mode=1, shard_rank=rank, shard_world=world, device=0)
        grp = HostConsumerGroup(eng, dist, eng.n, eng.m, eng.nnz_jac)
        xl, xu, _, _ = eng.get_bounds_info()
        xs = [problems.seeded_iterate(eng.get_starting_point(), xl, xu, 30 + i, mode) for i in range(4)]
        if rank == 0:
            for i in range(4):
                grp.x[i][:] = xs[i]
            grp.g[:] = np.nan
            grp.values[:] = np.nan
        dist.barrier()
        ref = NLPEngine(prob, device=0) if rank == 0 else None
        for k in (0, 1, 2, 3, 1, 1, 0):
            grp.step(k)
            if rank == 0:
                g_ref, v_ref = ref.eval_pair(xs[k])
                assert np.array_equal(np.array(grp.g), g_ref), "g assembled by the ranks differs"
                assert np.array_equal(np.array(grp.values), v_ref), "values assembled by the ranks differ"
        sent, total = eng.get_option("delta_sent_runs"), eng.get_option("delta_total_runs")
        assert 0 < total and 0 < sent < 7 * total, (sent, total)     # seven deliveries, the later ones partial
        # an error on ONE rank (a NaN among the nodes only the last rank evaluates -> RPM_E_NONFINITE there) reaches rank 0 as
        # an exception instead of a hang, and the group stays in step: the next step() is complete and correct again
        if rank == 0:
            grp.x[0][eng.n - 3] = np.nan                             # last control value of the last phase
        dist.barrier()
        try:
            grp.step(0)
            raised = False
        except Exception:
            raised = True
        flags = [None] * world
        dist.all_gather_object(flags, raised)
        assert flags[0] and flags[world - 1], flags
        if rank == 0:
            grp.x[0][:] = xs[0]
        dist.barrier()
        grp.step(0)
        if rank == 0:
            g_ref, v_ref = ref.eval_pair(xs[0])
            assert np.array_equal(np.array(grp.g), g_ref) and np.array_equal(np.array(grp.values), v_ref), "after a failed step"
        if ref is not None:
            ref.close()
        grp.close()      # sets pin_host = 0: the engine lets go of its registrations of the segment BEFORE it is unmapped
        eng.close()
    # ---- SweepShard around the real device solver ----
    opts = Options()
    opts.SetStringValue("hessian-approximation", "exact")
    qp = problems.quadrotor(2, 4)
    total = 5
    probe = NLPEngine(qp, opts, device=0)
    x0 = probe.get_starting_point()
    probe.close()
    rng = np.random.RandomState(0)
    starts = x0[None, :] * (1 + 1e-2 * rng.uniform(-1, 1, size=(total, x0.size)))

    def solve(start, count):
        e = NLPEngine(qp, opts, n_instances=count, device=0)
        s = BatchedIPM(e)
        r = s.solve(starts[start:start + count])
        s.close()
        e.close()
        return {"obj": r["obj"], "status": r["status"], "iterations": r["iterations"]}

    shard = SweepShard(dist, total)
    whole = shard.gather(solve(shard.start, shard.count))
    if rank == 0:
        serial = solve(0, total)
        assert np.array_equal(whole["status"], serial["status"].astype(np.float64)) and (whole["status"] == 0).all()
        assert np.array_equal(whole["iterations"], serial["iterations"].astype(np.float64))
        assert np.allclose(whole["obj"], serial["obj"], rtol=0, atol=1e-9)
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
