"""Worker of tests/test_dist_gloo.py (world_size 2, gloo, CPU).  Exercises the interval-sharding
partition / pack / all-gather / unpack logic of lpopc_amd.dist with the CPU oracle standing in for
the per-rank device evaluation, and the max-over-ranks timing reduction bench.py uses."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lpopc_amd import problems  # noqa: E402
from lpopc_amd.dist import SweepShard, pack_host, shard_instances, unpack_host  # noqa: E402
from lpopc_amd.engine import NLPEngine  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    prob = problems.launch(6, 5)
    eng = NLPEngine(prob, shard_mode=1, shard_rank=rank, shard_world=world)
    orc = Oracle(prob)
    xl, xu, _, _ = eng.get_bounds_info()
    x = problems.seeded_iterate(eng.get_starting_point(), xl, xu, 3)
    full = {0: orc.eval_g(x), 1: orc.eval_jac_g(x)}
    for which in (0, 1):
        all_segs = [eng.shard_segments(which, r)[0] for r in range(world)]
        stride = max(eng.shard_segments(which, r)[1] for r in range(world))
        # what this rank's GPU would have produced: only its own runs are valid, the rest is poison
        mine = np.full_like(full[which], np.nan)
        for off, ln, _ in all_segs[rank]:
            mine[off:off + ln] = full[which][off:off + ln]
        send = np.zeros(stride)
        packed = pack_host(mine, all_segs[rank])
        send[:packed.size] = packed
        recv = [torch.zeros(stride, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(recv, torch.from_numpy(send))
        gathered = torch.cat(recv).numpy()
        out = unpack_host(gathered, stride, all_segs, np.full_like(full[which], np.nan))
        assert np.array_equal(out, full[which]), "gathered vector differs from the single-rank result"
    # ONE collective for g + values of several instances on a RAGGED hp mesh (rpm_peer.hip's slot layout, host mirror):
    # pack this rank's runs of both vectors into its slot, all-gather the [world][slot] buffer in place, scatter the other
    # rank's slot — the assembled vectors equal the single-rank result for every instance
    from lpopc_amd.dist import pack_all_host, slot_layout, unpack_all_host
    rp = problems.launch()
    meshes = [([-1, -0.6, 0.1, 1], [5, 23, 2]), ([-1, 0.5, 1], [16, 17]), ([-1, 1], [33]), ([-1, -0.9, -0.5, 0.0, 0.25, 1], [3, 4, 7, 12, 16])]
    for i, (mesh, nodes) in enumerate(meshes):
        problems.set_mesh(rp.GetPhase(i), mesh, nodes)
    B = 3
    re_ = NLPEngine(rp, n_instances=B, shard_mode=1, shard_rank=rank, shard_world=world)
    ro = Oracle(rp)
    rxl, rxu, _, _ = re_.get_bounds_info()
    rx0 = re_.get_starting_point()[:re_.n]
    rxs = [problems.seeded_iterate(rx0, rxl, rxu, 40 + b) for b in range(B)]
    g_full = np.concatenate([ro.eval_g(x_) for x_ in rxs])
    v_full = np.concatenate([ro.eval_jac_g(x_) for x_ in rxs])
    layout, slot = slot_layout(re_, world, B)
    assert slot == re_.shard_slot_len()
    g_mine, v_mine = np.full_like(g_full, np.nan), np.full_like(v_full, np.nan)
    for b in range(B):
        for off, ln, _ in layout[rank][0]:
            g_mine[b * re_.m + off:b * re_.m + off + ln] = g_full[b * re_.m + off:b * re_.m + off + ln]
        for off, ln, _ in layout[rank][1]:
            v_mine[b * re_.nnz_jac + off:b * re_.nnz_jac + off + ln] = v_full[b * re_.nnz_jac + off:b * re_.nnz_jac + off + ln]
    buf = torch.zeros(world * slot, dtype=torch.float64)
    mine = buf[rank * slot:(rank + 1) * slot]
    mine.copy_(torch.from_numpy(pack_all_host(g_mine, v_mine, layout[rank], B, re_.m, re_.nnz_jac, slot)))
    dist.all_gather_into_tensor(buf, mine.clone())
    unpack_all_host(buf.numpy(), layout, B, re_.m, re_.nnz_jac, slot, g_mine, v_mine, skip_rank=rank)
    assert np.array_equal(g_mine, g_full) and np.array_equal(v_mine, v_full), "packed exchange differs from the single-rank result"
    # every entry of g / values is owned by exactly one rank
    cover_g, cover_v = np.zeros(re_.m, dtype=int), np.zeros(re_.nnz_jac, dtype=int)
    for r in range(world):
        for off, ln, _ in layout[r][0]:
            cover_g[off:off + ln] += 1
        for off, ln, _ in layout[r][1]:
            cover_v[off:off + ln] += 1
    assert (cover_g == 1).all() and (cover_v == 1).all()
    # bench.py's timing reduction: MAX over ranks
    t = torch.tensor([1.0 + rank, 10.0 - rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.tolist() == [float(world), 10.0]
    # weak scaling (independent instances): rank-specific iterates, nothing to exchange but the count
    cnt = torch.tensor([7.0])
    dist.all_reduce(cnt)
    assert cnt.item() == 7.0 * world
    # a sweep of independent NLP solves (row f-2) sharded by instance: every rank solves its share — here with the CPU
    # restatement standing in for the device solver — and only the per-instance verdicts are gathered
    from lpopc_amd.problem import Options
    from oracle import ipm_oracle
    assert [shard_instances(5, r, 2) for r in range(2)] == [(0, 3), (3, 2)] and shard_instances(4, 1, 4) == (1, 1)
    opts = Options()
    opts.SetStringValue("hessian-approximation", "exact")
    qp = problems.quadrotor(1, 3)
    qo = Oracle(qp, opts)
    total = 5
    rng = np.random.RandomState(0)
    starts = qo.starting_point()[None, :] * (1 + 1e-2 * rng.uniform(-1, 1, size=(total, qo.n)))

    def solve_local(start, count):
        rs = [ipm_oracle.solve(qo, starts[start + k]) for k in range(count)]
        return {"obj": [r["obj"] for r in rs], "status": [r["status"] for r in rs], "iterations": [r["iterations"] for r in rs]}

    shard = SweepShard(dist, total)
    whole = shard.gather(solve_local(shard.start, shard.count))
    serial = solve_local(0, total)
    for k in ("obj", "status", "iterations"):
        assert np.array_equal(whole[k], np.asarray(serial[k], dtype=np.float64)), k
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
