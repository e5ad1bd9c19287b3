"""Batched device entry points (eval_f_dev, eval_grad_f_dev, eval_h_dev with n_instances > 1) against one-instance engines, bit for bit.
Run on the GPU box: python tools/check_batched_api.py"""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine
from lpopc_amd.problem import Options
opts = Options(); opts.SetStringValue("hessian-approximation", "exact")
for name, mk in (("launch", lambda: problems.launch(3, 6)), ("quadrotor", lambda: problems.quadrotor(4, 5)), ("bd", problems.bryson_denham)):
    prob = mk()
    B = 5
    one = NLPEngine(prob, opts, device=0)
    many = NLPEngine(prob, opts, n_instances=B, device=0)
    xl, xu, _, _ = one.get_bounds_info()
    xs = np.stack([problems.seeded_iterate(one.get_starting_point(), xl, xu, 7 + i) for i in range(B)])
    lam = np.random.RandomState(0).uniform(-1, 1, (B, one.m))
    dx = torch.from_numpy(xs).cuda(); dl = torch.from_numpy(lam).cuda()
    dobj = torch.empty(B, dtype=torch.float64, device="cuda"); dgr = torch.empty((B, one.n), dtype=torch.float64, device="cuda")
    many.eval_f_dev(dx, dobj); many.eval_grad_f_dev(dx, dgr)
    i, j = one.eval_h_structure(); i2, j2 = many.eval_h_structure()
    dh = torch.empty((B, i.size), dtype=torch.float64, device="cuda")
    try:
        many.eval_h_dev(dx, 0.7, dl, dh)
        torch.cuda.synchronize()
        hok = all(np.array_equal(dh[b].cpu().numpy(), one.eval_h(xs[b], 0.7, lam[b])) for b in range(B))
    except Exception as e:
        hok = "ERR " + str(e)[:80]
    torch.cuda.synchronize()
    fok = all(dobj[b].item() == float(np.ravel(one.eval_f(xs[b]))[0]) for b in range(B))
    gok = all(np.array_equal(dgr[b].cpu().numpy(), one.eval_grad_f(xs[b])) for b in range(B))
    print(name, "f", fok, "grad", gok, "hess", hok, "struct", np.array_equal(i, i2))
    one.close(); many.close()
