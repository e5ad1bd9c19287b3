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
