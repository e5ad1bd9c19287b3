"""Multi-GPU data paths rehearsed on ONE GPU (SURVEY §4 "fake rank" mode; the 8-GPU box belongs to the driver):
  * 8 fake ranks on config 4's ragged hp mesh and on the 4-phase launch problem, several instances, with the one-role,
    role-looped and pipelined layouts: packed slots (rpm_shard_pack_all_dev) concatenated as the in-place all-gather
    would, scattered back (rpm_shard_unpack_all_dev) == the unsharded vectors, bit for bit;
  * the real RCCL collective at world size 1, captured in a hipGraph together with the tile kernel, pack and unpack;
  * two processes on the one GPU (gloo for control): the host-consumer mode (every rank stores its runs into one shared
    page-locked host array) and SweepShard around the real device interior-point solver."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


CASES = [
    ("config4_hp_mesh_role_looped", lambda: problems.config("hypersensitive"), "uniform", 1, 0, 3),
    ("config4_hp_mesh_pipelined", lambda: problems.config("hypersensitive"), "uniform", 1, 1, 20),
    ("config4_hp_mesh_one_role", lambda: problems.config("hypersensitive"), "uniform", 0, 0, 1),
    ("launch_16x8_pipelined", lambda: problems.launch(16, 8), "perturb", 1, 1, 5),
    ("launch_metric_one_role", lambda: problems.launch(64, 16), "perturb", 0, 0, 1),
]


@pytest.mark.parametrize("name,make,mode,role_loop,pipeline,B", CASES, ids=[c[0] for c in CASES])
def test_eight_fake_ranks_packed_exchange(built, name, make, mode, role_loop, pipeline, B):
    import torch
    prob = make()
    world = 8
    ref = NLPEngine(prob, n_instances=B, device=0, role_loop=role_loop)
    ref.set_option("pipeline", pipeline)
    ref.set_option("instance_align", 16)
    xl, xu, _, _ = ref.get_bounds_info()
    x0 = ref.get_starting_point()[:ref.n]
    xs = np.stack([problems.seeded_iterate(x0, xl, xu, 3 + b, mode) for b in range(B)])
    dx = torch.from_numpy(xs).cuda()
    sg, sv = ref.get_option("stride_g"), ref.get_option("stride_values")
    g_ref = torch.full((B, sg), np.nan, dtype=torch.float64, device="cuda")
    v_ref = torch.full((B, sv), np.nan, dtype=torch.float64, device="cuda")
    ref.eval_pair_dev(dx, g_ref, v_ref)
    engs = []
    for r in range(world):
        e = NLPEngine(prob, n_instances=B, shard_mode=1, shard_rank=r, shard_world=world, device=0, role_loop=role_loop)
        e.set_option("pipeline", pipeline)
        e.set_option("instance_align", 16)
        engs.append(e)
    slot = engs[0].shard_slot_len()
    assert all(e.shard_slot_len() == slot for e in engs) and slot % 16 == 0
    gathered = torch.full((world * slot,), np.nan, dtype=torch.float64, device="cuda")
    own = []
    for r, e in enumerate(engs):
        dg = torch.full((B, sg), np.nan, dtype=torch.float64, device="cuda")
        dv = torch.full((B, sv), np.nan, dtype=torch.float64, device="cuda")
        e.eval_pair_dev(dx, dg, dv)
        e.shard_pack_all_dev(dg, dv, gathered[r * slot:(r + 1) * slot])
        own.append((dg, dv))
    torch.cuda.synchronize()
    assert ref.get_option("pipeline_active") == pipeline
    # rank 5 ends the step: its own runs are in place, the others arrive through the gathered buffer
    dg, dv = own[5]
    engs[5].shard_unpack_all_dev(gathered, dg, dv, skip_own=True)
    # and a rank that scatters everything, its own slot included
    og = torch.full((B, sg), np.nan, dtype=torch.float64, device="cuda")
    ov = torch.full((B, sv), np.nan, dtype=torch.float64, device="cuda")
    engs[2].shard_unpack_all_dev(gathered, og, ov, skip_own=False)
    torch.cuda.synchronize()
    m, nnz = ref.m, ref.nnz_jac
    for a_g, a_v in ((dg, dv), (og, ov)):
        assert torch.equal(a_g[:, :m], g_ref[:, :m]) and torch.equal(a_v[:, :nnz], v_ref[:, :nnz])
    assert not torch.isnan(g_ref[:, :m]).any() and not torch.isnan(v_ref[:, :nnz]).any()
    for e in engs + [ref]:
        e.close()


def test_rccl_all_gather_captured_in_the_step_graph(built):
    """The real collective (RCCL, world size 1 — the only size one GPU allows) inside a hipGraph with the tile kernel, the
    pack and the unpack kernel: what bench.py's strong-scaling sections replay.  Runs in a process of its own: a process
    group and graphs that hold collective nodes are process-wide state that the test session should not inherit."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_gpu_rccl_graph_worker.py")], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-5000:]
    assert "rccl graph ok" in r.stdout


def test_two_processes_on_one_gpu_host_consumer_and_sweep(built):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_gpu_two_rank_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-5000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout
