"""Multi-GPU glue (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

Two ways the path shards:
  * independent instances (multi-start / MPC sweep): every rank owns whole problems; no collective
    on the data path (bench.py default, weak scaling);
  * mesh intervals of ONE problem: rank r evaluates a contiguous run of each phase's tiles; its share
    of g / values is a list of contiguous runs (rpm_shard_segments).  `IntervalExchange` packs the runs of both
    vectors (all instances) into one slot, all-gathers ONE buffer over xGMI and scatters the other ranks' slots back
    into TNLP order (device consumer); `HostConsumerGroup` lets every rank store its runs straight into one shared
    page-locked host array over its own PCIe link (host consumer, e.g. Ipopt in rank 0's process).
    `IntervalGather` is the round-1 two-collective form, kept for comparison.
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


# ---- one packed slot per rank: g AND values of all instances in ONE collective -------------------------------------------------
def slot_layout(eng, world, n_instances=1):
    """Host mirror of rpm_peer.hip's slot layout: per rank (segments_g, segments_values, packed_len_g, packed_len_values), and
    the slot length (doubles, the same for every rank, whole 128-byte lines)."""
    per = []
    for r in range(world):
        sg, pg = eng.shard_segments(0, r)
        sv, pv = eng.shard_segments(1, r)
        per.append((sg, sv, pg, pv))
    slot = max(pg + pv for _, _, pg, pv in per) * n_instances
    return per, (slot + 15) // 16 * 16


def pack_all_host(g, values, layout_r, n_instances, stride_g, stride_values, slot):
    """Host mirror of rpm_shard_pack_all_dev for one rank."""
    sg, sv, pg, pv = layout_r
    out = np.zeros(slot, dtype=np.float64)
    for b in range(n_instances):
        base = b * (pg + pv)
        for off, ln, pos in sg:
            out[base + pos:base + pos + ln] = g[b * stride_g + off:b * stride_g + off + ln]
        for off, ln, pos in sv:
            out[base + pg + pos:base + pg + pos + ln] = values[b * stride_values + off:b * stride_values + off + ln]
    return out


def unpack_all_host(gathered, layout, n_instances, stride_g, stride_values, slot, g, values, skip_rank=None):
    """Host mirror of rpm_shard_unpack_all_dev."""
    for r, (sg, sv, pg, pv) in enumerate(layout):
        if r == skip_rank:
            continue
        for b in range(n_instances):
            base = r * slot + b * (pg + pv)
            for off, ln, pos in sg:
                g[b * stride_g + off:b * stride_g + off + ln] = gathered[base + pos:base + pos + ln]
            for off, ln, pos in sv:
                values[b * stride_values + off:b * stride_values + off + ln] = gathered[base + pg + pos:base + pg + pos + ln]
    return g, values


class IntervalExchange:
    """Interval-sharded step on the GPU: this rank's runs of g and of the Jacobian values (all instances) are packed into its
    slot of ONE [world][slot] buffer, the buffer is all-gathered in place (RCCL over xGMI: one collective per step), the
    other ranks' slots are scattered into TNLP order.  pack -> all_gather -> unpack are plain stream operations, so the
    caller may capture a whole step (tile kernel included) in a hipGraph."""

    def __init__(self, eng, dist, world, rank):
        import torch
        self.eng, self.dist, self.world, self.rank = eng, dist, world, rank
        self.slot = eng.shard_slot_len()
        self.buf = torch.zeros(world * self.slot, dtype=torch.float64, device="cuda")
        self.mine = self.buf[rank * self.slot:(rank + 1) * self.slot]    # in-place all-gather: send = own slot of recv

    def exchange(self, d_g, d_values):
        self.eng.shard_pack_all_dev(d_g, d_values, self.mine)
        self.dist.all_gather_into_tensor(self.buf, self.mine)
        self.eng.shard_unpack_all_dev(self.buf, d_g, d_values, skip_own=True)

    def bytes_received_per_step(self):
        return (self.world - 1) * self.slot * 8


# ---- host consumer: every rank stores its own runs straight into ONE page-locked host array over ITS OWN PCIe link ----------------
class HostConsumerGroup:
    """SURVEY §5 plan (c): when the consumer of g / values is a host-side solver (Ipopt lives in rank 0's process), no
    GPU-to-GPU gather is needed — x, g and values live in ONE shared host segment that every rank maps and page-locks; rank
    r's interval-sharded engine reads x from it and stores ITS runs of g and (by difference, option delta_values) of the
    Jacobian values into it over its own PCIe link.  Control is two words per rank in the same segment: rank 0 publishes a
    sequence number, every rank evaluates and publishes "done"; no collective on the data path.

    The segment is POSIX shared memory (multiprocessing.shared_memory); `dist` is only used to hand its name round."""

    def __init__(self, eng, dist, n, m, nnz, n_instances=1, n_x_slots=4, timeout_s=60.0):
        from multiprocessing import shared_memory
        self.timeout_s = float(timeout_s)
        self.eng, self.dist = eng, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        B = n_instances
        self.counts = (n_x_slots * B * n, B * m, B * nnz)
        ctrl = 64 * (self.world + 1)                       # one cache line per word
        nbytes = ctrl * 8 + 8 * sum(self.counts) + (n_x_slots + 3) * 4096
        name = [None]
        if self.rank == 0:
            self.shm = shared_memory.SharedMemory(create=True, size=nbytes)
            name[0] = self.shm.name
        dist.broadcast_object_list(name, src=0)
        if self.rank != 0:
            self.shm = shared_memory.SharedMemory(name=name[0])
            try:                                            # the creator unlinks; keep the tracker of this process out of it
                from multiprocessing import resource_tracker
                resource_tracker.unregister(self.shm._name, "shared_memory")
            except Exception:
                pass
        buf = np.frombuffer(self.shm.buf, dtype=np.float64)
        self.ctrl = np.frombuffer(self.shm.buf, dtype=np.int64, count=ctrl)
        pos = ctrl

        def take(count):
            nonlocal pos
            pos = (pos + 511) // 512 * 512                  # page-aligned arrays
            a = buf[pos:pos + count]
            pos += count
            return a
        self.x = [take(B * n) for _ in range(n_x_slots)]    # each on its own pages: registered one by one
        self.g = take(self.counts[1])
        self.values = take(self.counts[2])
        if self.rank == 0:
            self.ctrl[:] = 0
        dist.barrier()
        eng.set_option("pin_host", 1)
        eng.set_option("delta_values", 1)
        self.seq = 0

    def _wait(self, word, deadline, what):
        """Poll a control word until it reaches +seq (ok) or -seq (the peer failed), with a deadline: an error on one rank — e.g.
        RPM_E_NONFINITE on the rank whose intervals hold the NaN — must surface on every rank, not hang the others at 100 % CPU."""
        import time
        spins = 0
        while True:
            v = int(self.ctrl[word])
            if v >= self.seq:
                return
            if v == -self.seq:
                raise RuntimeError("HostConsumerGroup: %s reported a failed evaluation in step %d" % (what, self.seq))
            spins += 1
            if spins > 2000:                                   # ~ the first 100 us spin, then yield the core
                if time.monotonic() > deadline:
                    raise TimeoutError("HostConsumerGroup: no answer from %s in step %d within %.0f s" % (what, self.seq, self.timeout_s))
                time.sleep(50e-6)

    def step(self, x_slot):
        """One (eval_g, eval_jac_g) pair of the iterate in x slot `x_slot`, all ranks together; returns when the whole g and
        values arrays are complete (on rank 0: for the consumer; the other ranks return after their own share).  A rank whose
        evaluation fails publishes -seq; every rank then raises (rank 0 after it has heard from all ranks, so the group stays
        in step and the next step() can proceed)."""
        import time
        self.seq += 1
        deadline = time.monotonic() + self.timeout_s
        if self.rank == 0:
            self.ctrl[0] = self.seq                          # go
        else:
            self._wait(0, deadline, "rank 0 (go)")
        err = None
        try:
            self.eng.eval_pair(self.x[x_slot], self.g, self.values)   # own rows / runs only (interval-sharded engine)
            self.ctrl[64 * (self.rank + 1)] = self.seq        # done
        except Exception as ex:                               # expected error returns included (RPM_E_NONFINITE ...)
            err = ex
            self.ctrl[64 * (self.rank + 1)] = -self.seq       # failed
        if self.rank == 0:
            failed = []
            for r in range(1, self.world):
                try:
                    self._wait(64 * (r + 1), deadline, "rank %d" % r)
                except RuntimeError:
                    failed.append(r)
            if err is None and failed:
                err = RuntimeError("HostConsumerGroup: evaluation failed on rank(s) %s in step %d" % (failed, self.seq))
        if err is not None:
            raise err

    def close(self):
        try:
            self.eng.set_option("pin_host", 0)      # releases the page-locked registrations of the segment BEFORE it is unmapped
        except Exception:
            pass
        self.x = self.g = self.values = self.ctrl = None
        try:
            self.dist.barrier()
        except Exception:
            pass
        try:
            self.shm.close()
            if self.rank == 0:
                self.shm.unlink()
        except Exception:
            pass


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
