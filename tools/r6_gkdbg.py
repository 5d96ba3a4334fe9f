import json, os, sys
ROOT = "/root/repo"
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")):
    sys.path.insert(0, p)
import numpy as np, torch.distributed as dist, cfmm
from cfmm import synthetic, problem as PM
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
net = synthetic.make_network(200, m_cp2=20000, m_gk_sum=1000, seed=3)
p = cfmm.distributed.sharded_problem(net, cfmm.Arbitrage(net["c"]), dist=dist, device=0, allreduce="oneshot", rccl=False)
orig = PM.Problem._run
leg = [0]
def traced(ctx, nu, total, **kw):
    st = orig(ctx, nu, total, **kw)
    nu1, psi1 = ctx.get_solution()
    box = [None] * world
    dist.all_gather_object(box, (None if nu is None else np.asarray(nu).tolist(), nu1.tolist(), psi1.tolist(), st["evals"], st["status"]))
    if rank == 0:
        a, b = box
        print("leg", leg[0], "pg_rule", kw.get("pg_rule", 0), "evals", a[3], b[3], "status", a[4], b[4], "start same", a[0] == b[0], "nu same", a[1] == b[1], "psi same", a[2] == b[2],
              "ndiff nu", int((np.array(a[1]) != np.array(b[1])).sum()), flush=True)
    leg[0] += 1
    return st
PM.Problem._run = staticmethod(traced)
v = p.solve(tol=1e-6, max_evals=1500, method="lbfgs")
dist.barrier(); p.close(); dist.destroy_process_group()
