"""TEST INFRASTRUCTURE -- ctypes binding of oracle/cfmm_oracle.c (built by oracle/Makefile).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcfmm_oracle.so")

K_CP2, K_W2, K_SUM2, K_CURVE2, K_POW2 = 0, 1, 2, 3, 4
KIND2 = dict(cp2=K_CP2, w2=K_W2, sum2=K_SUM2, curve2=K_CURVE2, pow2=K_POW2)


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "cfmm_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class Opts(C.Structure):
    _fields_ = [("tol_gap", C.c_double), ("tol_infeas", C.c_double), ("armijo", C.c_double),
                ("max_step", C.c_double), ("max_evals", C.c_int), ("memory", C.c_int),
                ("pg_rule", C.c_int), ("pad", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("evals", C.c_int), ("iters", C.c_int), ("status", C.c_int),
                ("dual_value", C.c_double), ("primal_value", C.c_double), ("gap", C.c_double),
                ("infeas", C.c_double), ("seconds", C.c_double), ("pg", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p
        L.oracle_create.restype = vp; L.oracle_create.argtypes = [C.c_int]
        L.oracle_destroy.argtypes = [vp]
        L.oracle_set_threads.argtypes = [vp, C.c_int]
        L.oracle_add_pools2.argtypes = [vp, C.c_int, C.c_int64, dp, dp, dp, dp, ip, ip, ip]
        L.oracle_add_poolsN.argtypes = [vp, C.c_int, C.c_int64, ip, dp, dp, dp]
        L.oracle_clear_pools.argtypes = [vp]
        L.oracle_set_utility.argtypes = [vp, dp, dp, ip]
        L.oracle_set_ties.argtypes = [vp, C.c_int, ip, dp]
        L.oracle_eval.restype = C.c_double; L.oracle_eval.argtypes = [vp, dp, dp, dp]
        L.oracle_trades2.argtypes = [vp, C.c_int, dp, dp, dp]
        L.oracle_tradesN.argtypes = [vp, C.c_int, dp, dp]
        L.oracle_start.restype = None; L.oracle_start.argtypes = [vp, dp, C.c_int]
        L.oracle_step.restype = C.c_int; L.oracle_step.argtypes = [vp, C.c_double, dp, dp, C.POINTER(Opts)]
        L.oracle_get.restype = None; L.oracle_get.argtypes = [vp, dp, dp, C.POINTER(Stats)]
        L.oracle_get_psi.restype = None; L.oracle_get_psi.argtypes = [vp, dp]
        L.oracle_solve.restype = C.c_int
        L.oracle_solve.argtypes = [vp, dp, C.POINTER(Opts), C.POINTER(Stats), dp, dp]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


class Oracle:
    """Holds a bucketed network (the dict layout of cfmm.synthetic / cfmm.problem.pack)."""

    def __init__(self, n_tokens, threads=1):
        self.n = int(n_tokens)
        self.L = lib()
        self.h = self.L.oracle_create(self.n)
        self.L.oracle_set_threads(self.h, threads)
        self._keep = []
        self.b2 = []      # (name, dict)
        self.bn = []      # (k, dict)

    def __del__(self):
        try:
            self.L.oracle_destroy(self.h)
        except Exception:
            pass

    def add_pools2(self, kind, Ra, Rb, fee, ia, ib, param=None, tied=None):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        g = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        Ra, Rb, fee, ia, ib = f(Ra), f(Rb), f(fee), g(ia), g(ib)
        param = f(param) if param is not None else None
        tied = g(tied) if tied is not None else None
        self._keep += [Ra, Rb, fee, ia, ib, param, tied]
        rc = self.L.oracle_add_pools2(self.h, KIND2[kind], len(Ra), _d(Ra), _d(Rb), _d(fee), _d(param), _i(ia), _i(ib), _i(tied))
        assert rc == 0
        self.b2.append((kind, dict(Ra=Ra, Rb=Rb, fee=fee, ia=ia, ib=ib, param=param, tied=tied)))

    def add_poolsN(self, idx, R, w, fee):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        R = np.ascontiguousarray(R, dtype=np.float64); w = np.ascontiguousarray(w, dtype=np.float64)
        fee = np.ascontiguousarray(fee, dtype=np.float64)
        k, m = R.shape
        self._keep += [idx, R, w, fee]
        rc = self.L.oracle_add_poolsN(self.h, k, m, _i(idx), _d(R), _d(w), _d(fee))
        assert rc == 0
        self.bn.append((k, dict(idx=idx, R=R, w=w, fee=fee)))

    def add_network(self, net):
        """net: bucket dict as produced by cfmm.synthetic.make_network / cfmm.problem.pack"""
        if "cp2" in net:
            b = net["cp2"]; self.add_pools2("cp2", b["Ra"], b["Rb"], b["fee"], b["ia"], b["ib"])
        if "w2" in net:
            b = net["w2"]; self.add_pools2("w2", b["Ra"], b["Rb"], b["fee"], b["ia"], b["ib"], param=b["wa"])
        if "sum2" in net:
            b = net["sum2"]; self.add_pools2("sum2", b["Ra"], b["Rb"], b["fee"], b["ia"], b["ib"], tied=b.get("tied"))
        if "curve2" in net:
            b = net["curve2"]; self.add_pools2("curve2", b["Ra"], b["Rb"], b["fee"], b["ia"], b["ib"], param=b["alpha"])
        if "pow2" in net:
            b = net["pow2"]; self.add_pools2("pow2", b["Ra"], b["Rb"], b["fee"], b["ia"], b["ib"], param=b["t"])
        for k in sorted(net.get("gn", {})):
            b = net["gn"][k]; self.add_poolsN(b["idx"], b["R"], b["w"], b["fee"])

    def set_utility(self, c, h=None, ctype=None):
        c = np.ascontiguousarray(c, dtype=np.float64)
        h = np.zeros(self.n) if h is None else np.ascontiguousarray(h, dtype=np.float64)
        ctype = np.zeros(self.n, dtype=np.int32) if ctype is None else np.ascontiguousarray(ctype, dtype=np.int32)
        self.c, self.hh, self.ctype = c, h, ctype
        self.L.oracle_set_utility(self.h, _d(c), _d(h), _i(ctype))

    def set_ties(self, grp, off):
        grp = np.ascontiguousarray(grp, dtype=np.int32); off = np.ascontiguousarray(off, dtype=np.float64)
        self.L.oracle_set_ties(self.h, int(grp.max()) + 1, _i(grp), _d(off))

    def eval(self, nu, want_diag=False):
        nu = np.ascontiguousarray(nu, dtype=np.float64)
        psi = np.zeros(self.n); diag = np.zeros(self.n) if want_diag else None
        f = self.L.oracle_eval(self.h, _d(nu), _d(psi), _d(diag))
        return (f, psi, diag) if want_diag else (f, psi)

    def trades2(self, b, nu):
        nu = np.ascontiguousarray(nu, dtype=np.float64)
        m = len(self.b2[b][1]["Ra"])
        ya = np.zeros(m); yb = np.zeros(m)
        self.L.oracle_trades2(self.h, b, _d(nu), _d(ya), _d(yb))
        return ya, yb

    def tradesN(self, b, nu):
        nu = np.ascontiguousarray(nu, dtype=np.float64)
        k, d = self.bn[b]
        y = np.zeros_like(d["R"])
        self.L.oracle_tradesN(self.h, b, _d(nu), _d(y))
        return y

    def solve_sharded(self, nu0, allreduce, tol=1e-6, max_evals=2000, memory=0, armijo=1e-4, max_step=2.0, pg_rule=0):
        """the pool-sharded outer loop, exactly as the GPU library runs it: every rank evaluates ITS
        pools, `allreduce(vec)` sums [psi | sum arb | diag] over the ranks in place, and every rank
        takes the identical step.  `self` holds this rank's shard only."""
        nu0 = np.ascontiguousarray(nu0, dtype=np.float64)
        if memory == 0:
            memory = 8 if (self.n <= 32 or (getattr(self, 'ctype', None) is not None and (self.ctype >= 3).any())) else 3
        o = Opts(tol, tol, armijo, max_step, max_evals, memory, pg_rule, 0)
        n = self.n
        self.L.oracle_start(self.h, _d(nu0), memory)
        nu = np.zeros(n); buf = np.zeros(2 * n + 1)
        st = Stats()
        first, status = True, 0
        while status == 0:
            self.L.oracle_get(self.h, _d(nu), None, None)               # the trial prices
            psi = np.zeros(n); diag = np.zeros(n)
            f = self.L.oracle_eval(self.h, _d(nu), _d(psi), _d(diag) if first else None)
            buf[:n] = psi; buf[n] = f; buf[n + 1:] = diag if first else 0.0
            allreduce(buf)
            psi = np.ascontiguousarray(buf[:n]); diag = np.ascontiguousarray(buf[n + 1:])
            status = self.L.oracle_step(self.h, float(buf[n]), _d(psi), _d(diag), C.byref(o))
            first = False
        nu_acc = np.zeros(n); psi_acc = np.zeros(n)
        self.L.oracle_get(self.h, None, _d(nu_acc), C.byref(st))
        self.L.oracle_get_psi(self.h, _d(psi_acc))
        return dict(nu=nu_acc, psi=psi_acc, pg=st.pg, seconds=0.0, evals=st.evals, iters=st.iters, status=st.status, dual_value=st.dual_value,
                    primal_value=st.primal_value, gap=st.gap, infeas=st.infeas)

    def solve(self, nu0, tol=1e-6, max_evals=2000, memory=0, armijo=1e-4, max_step=2.0, pg_rule=0):
        nu0 = np.ascontiguousarray(nu0, dtype=np.float64)
        if memory == 0:                     # same auto rule as cfmm_solve
            memory = 8 if (self.n <= 32 or (getattr(self, 'ctype', None) is not None and (self.ctype >= 3).any())) else 3
        o = Opts(tol, tol, armijo, max_step, max_evals, memory, pg_rule, 0)
        st = Stats()
        nu = np.zeros(self.n); psi = np.zeros(self.n)
        self.L.oracle_solve(self.h, _d(nu0), C.byref(o), C.byref(st), _d(nu), _d(psi))
        return dict(nu=nu, psi=psi, evals=st.evals, iters=st.iters, status=st.status,
                    dual_value=st.dual_value, primal_value=st.primal_value, gap=st.gap,
                    infeas=st.infeas, seconds=st.seconds, pg=st.pg)
