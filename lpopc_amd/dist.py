"""Multi-GPU glue (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

Two ways the path shards:
  * independent instances (multi-start / MPC sweep): every rank owns whole problems; no collective
    on the data path (bench.py default, weak scaling);
  * mesh intervals of ONE problem: rank r evaluates a contiguous run of each phase's tiles; its share
    of g / values is a list of contiguous runs (rpm_shard_segments).  `IntervalGather` packs those runs,
    all-gathers the packed buffers over xGMI and scatters them back into TNLP order on every rank.
The reference has no counterpart (lpopc is single-process).
"""
import numpy as np


def pack_host(full, segments):
    """Host mirror of rpm_shard_pack_dev (used by the CPU gloo tests)."""
    out = np.zeros(sum(l for _, l, _ in segments), dtype=full.dtype)
    for off, ln, pos in segments:
        out[pos:pos + ln] = full[off:off + ln]
    return out


def unpack_host(gathered, stride, all_segments, full):
    """Host mirror of rpm_shard_unpack_dev: gathered is [world*stride]."""
    for r, segments in enumerate(all_segments):
        for off, ln, pos in segments:
            full[off:off + ln] = gathered[r * stride + pos:r * stride + pos + ln]
    return full


class IntervalGather:
    """All-gather of the interval-sharded g / values segments on the GPU."""

    def __init__(self, eng, dist, world):
        import torch
        self.eng, self.dist, self.world = eng, dist, world
        self.stride, self.send, self.recv = [], [], []
        for which in (0, 1):
            stride = max(eng.shard_segments(which, r)[1] for r in range(world))
            self.stride.append(stride)
            self.send.append(torch.zeros(stride, dtype=torch.float64, device="cuda"))
            self.recv.append(torch.zeros(world * stride, dtype=torch.float64, device="cuda"))

    def all_gather(self, d_g, d_values):
        for which, full in ((0, d_g), (1, d_values)):
            if full is None:
                continue
            self.eng.shard_pack_dev(which, full, self.send[which])
            self.dist.all_gather_into_tensor(self.recv[which], self.send[which])
            self.eng.shard_unpack_dev(which, self.recv[which], self.stride[which], full)


# ---- sweeps of independent instances (bench.py default; the device NLP solver rpm_ipm_*) -------------------------------------
def shard_instances(total, rank, world):
    """Contiguous share [start, start + count) of `total` independent instances for `rank`: sizes differ by at most one,
    nothing is exchanged on the data path."""
    base, extra = divmod(int(total), int(world))
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


class SweepShard:
    """One rank's part of a sweep: solve its instances with `solve_local(start, count) -> dict of per-instance numpy arrays`
    (on a GPU: a BatchedIPM over an engine with `count` instances), then gather the per-instance verdicts — a few numbers
    per instance, not iterates — so that every rank can report on the whole sweep.  The only collective is this gather."""

    def __init__(self, dist, total):
        self.dist, self.total = dist, int(total)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.start, self.count = shard_instances(self.total, self.rank, self.world)

    def gather(self, local, keys=("obj", "status", "iterations")):
        import torch
        out = {}
        longest = shard_instances(self.total, 0, self.world)[1]
        # RCCL ("nccl") moves device memory only: stage the few verdict numbers on this rank's GPU; gloo takes host tensors
        dev = torch.device("cuda", torch.cuda.current_device()) if self.dist.get_backend() == "nccl" else torch.device("cpu")
        send = torch.zeros(len(keys) * longest, dtype=torch.float64)
        for j, k in enumerate(keys):
            a = np.asarray(local[k], dtype=np.float64).reshape(-1)
            assert a.size == self.count, "solve_local returned %d entries for %d instances" % (a.size, self.count)
            send[j * longest:j * longest + a.size] = torch.from_numpy(a)
        send = send.to(dev)
        recv = torch.zeros(self.world * len(keys) * longest, dtype=torch.float64, device=dev)
        self.dist.all_gather_into_tensor(recv, send)    # ONE collective for all keys
        recv = recv.cpu().numpy().reshape(self.world, len(keys), longest)
        for j, k in enumerate(keys):
            out[k] = np.concatenate([recv[r, j, :shard_instances(self.total, r, self.world)[1]] for r in range(self.world)])
        return out
