"""TEST-ONLY stand-in for cfmm._lib.Context backed by the C oracle, so that the host logic of
cfmm.problem (packing, start prices, kink ties, fill recovery) can be exercised without a GPU.
Never imported by the product."""
import numpy as np

from oracle.c_oracle import Oracle

_K = {0: "cp2", 1: "w2", 2: "sum2", 3: "curve2", 4: "pow2"}


class OracleContext:
    backend = "oracle:cpu"

    def __init__(self, n_tokens, threads=1):
        self.n = n_tokens
        self.threads = threads
        self.b2 = {}
        self.bn = {}
        self.flags = None
        self.util = None
        self.ties = None
        self._o = None
        self._nu = None
        self._psi = None

    def upload_pools2(self, kind, Ra, Rb, fee, ia, ib, param=None):
        self.b2[kind] = dict(Ra=Ra, Rb=Rb, fee=fee, ia=ia, ib=ib, param=param); self._o = None

    def upload_poolsN(self, idx, R, w, fee):
        self.bn[R.shape[0]] = dict(idx=idx, R=R, w=w, fee=fee); self._o = None

    def set_pool_flags(self, kind, flags):
        self.flags = None if flags is None else np.asarray(flags, dtype=np.int32).copy(); self._o = None

    def set_utility(self, c, h=None, ctype=None):
        self.util = (c, h, ctype); self._o = None

    def set_ties(self, grp=None, off=None):
        self.ties = None if grp is None else (grp, off); self._o = None

    def _build(self):
        if self._o is None:
            o = Oracle(self.n, threads=self.threads)
            self.order2 = []
            for kind, b in sorted(self.b2.items()):
                o.add_pools2(_K[kind], b["Ra"], b["Rb"], b["fee"], b["ia"], b["ib"], param=b["param"],
                             tied=self.flags if kind == 2 else None)
                self.order2.append(kind)
            self.ordern = []
            for k, b in sorted(self.bn.items()):
                o.add_poolsN(b["idx"], b["R"], b["w"], b["fee"]); self.ordern.append(k)
            o.set_utility(*self.util) if self.util else None
            if self.ties is not None:
                o.set_ties(*self.ties)
            self._o = o
        return self._o

    def eval_dual(self, nu, want_diag=False):
        return self._build().eval(nu, want_diag)

    def _need_utility(self):
        # as libcfmm_hip.so: cfmm_solve on a context that never received a utility is CFMM_E_STATE (cfmm_hip.hip: cfmm_solve)
        if self.util is None:
            from cfmm._lib import CfmmError
            raise CfmmError("cfmm_solve: cfmm_set_utility has not been called")

    def solve(self, nu0=None, tol=1e-6, max_evals=2000, memory=0, iters_per_graph=8, pg_rule=0, **kw):
        self._need_utility()
        o = self._build()
        r = o.solve(nu0 if nu0 is not None else self._nu, tol=tol, max_evals=max_evals, memory=memory, pg_rule=pg_rule)
        self._nu, self._psi = r["nu"], r["psi"]
        return dict(evals=r["evals"], iters=r["iters"], status=r["status"], n_ranks=1, dual_value=r["dual_value"],
                    primal_value=r["primal_value"], gap=r["gap"], infeas=r["infeas"], wall_seconds=r["seconds"],
                    device_seconds=r["seconds"], pg=r["pg"], pool_subproblems=0)

    def pool_count(self):
        return sum(len(b["Ra"]) for b in self.b2.values()) + sum(b["R"].shape[1] for b in self.bn.values())

    def get_nu(self):
        return self._nu.copy()

    def set_nu(self, nu):
        self._nu = np.asarray(nu, dtype=np.float64).copy()

    def get_psi(self):
        return self._psi.copy()

    def get_solution(self):
        return self._nu.copy(), self._psi.copy()

    def get_trades2(self, kind, m):
        o = self._build()
        ya, yb = o.trades2(self.order2.index(kind), self._nu)
        y = np.stack([ya, yb])
        return np.maximum(-y, 0.0), np.maximum(y, 0.0)

    def get_tradesN(self, k, m):
        o = self._build()
        y = o.tradesN(self.ordern.index(k), self._nu)
        return np.maximum(-y, 0.0), np.maximum(y, 0.0)

    def close(self):
        pass


class ShardedOracleContext(OracleContext):
    """TEST-ONLY: one rank of a pool-sharded job on CPU.  This rank's shard is evaluated by the C oracle and
    [psi | sum arb | diag] is all-reduced over torch.distributed (gloo) once per dual evaluation -- the structure
    libcfmm_hip.so runs with RCCL -- so that the product's host-side sharded control flow (cfmm.problem /
    cfmm.distributed: global decisions, start-price broadcast, all-gathered kink ties) runs under world_size > 1."""

    def __init__(self, n_tokens, dist):
        super().__init__(n_tokens)
        self.dist = dist
        self.allreduces = 0
        self.solves = 0
        self.start_prices = []

    def _allreduce(self, buf):
        import torch
        t = torch.from_numpy(buf)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        self.allreduces += 1

    def eval_dual(self, nu, want_diag=False):
        r = self._build().eval(nu, True)
        buf = np.concatenate([r[1], [r[0]], r[2]])
        self._allreduce(buf)
        n = self.n
        return (buf[n], buf[:n].copy(), buf[n + 1:].copy()) if want_diag else (buf[n], buf[:n].copy())

    def solve(self, nu0=None, tol=1e-6, max_evals=2000, memory=0, iters_per_graph=8, pg_rule=0, **kw):
        self._need_utility()
        o = self._build()
        nu0 = nu0 if nu0 is not None else self._nu
        self.start_prices.append(np.array(nu0, dtype=np.float64))
        self.solves += 1
        r = o.solve_sharded(nu0, self._allreduce, tol=tol, max_evals=max_evals, memory=memory, pg_rule=pg_rule)
        self._nu, self._psi = r["nu"], r["psi"]
        return dict(evals=r["evals"], iters=r["iters"], status=r["status"], n_ranks=self.dist.get_world_size(),
                    dual_value=r["dual_value"], primal_value=r["primal_value"], gap=r["gap"], infeas=r["infeas"],
                    wall_seconds=0.0, device_seconds=0.0, pg=r["pg"], pool_subproblems=0)
