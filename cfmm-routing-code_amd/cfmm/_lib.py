"""ctypes binding of libcfmm_hip.so (include/cfmm.h).  No CPU fallback: if the HIP extension
is not built, or no gfx950 device is visible, every entry point raises."""
import ctypes as C
import struct
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CFMM_LIB selects another build of the same library (kernel-tuning variants under cfmm/variants/)
_SO = os.environ.get("CFMM_LIB") or os.path.join(_HERE, "libcfmm_hip.so")
_CSRC = os.path.join(os.path.dirname(_HERE), "csrc")

POOL_CP2, POOL_W2, POOL_SUM2, POOL_CURVE2, POOL_POW2 = 0, 1, 2, 3, 4
TIME_ALL = 100
TIME_TABLE = 200      # the K-asset table's own launch (include/cfmm.h)
GE, EQ, FREE = 0, 1, 2
ULOG, UQUAD = 3, 4          # the utility table: u = c log(psi + h) / u = c psi - psi^2 / (2 h)   (include/cfmm.h)
MAX_POOL_SIZE = 8
POOLK = {"stable": 0, "sum": 1}     # the K-asset table's kinds (include/cfmm.h: CFMM_POOLK_*)
STATUS = {1: "optimal", 2: "stalled", 3: "max_evals"}


class CfmmError(RuntimeError):
    """an error return of libcfmm_hip; `code` is the library's return code (include/cfmm.h: CFMM_E_*), None for errors raised on the
    Python side of the binding"""
    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


E_ARG, E_HIP, E_STATE, E_LIMIT, E_RCCL, E_NUMERIC, E_UNSUPPORTED = -1, -2, -3, -4, -5, -6, -7      # include/cfmm.h


class Opts(C.Structure):
    _fields_ = [("tol_gap", C.c_double), ("tol_infeas", C.c_double), ("armijo", C.c_double),
                ("max_step", C.c_double), ("max_evals", C.c_int32), ("memory", C.c_int32),
                ("iters_per_graph", C.c_int32), ("pg_rule", C.c_int32),
                ("method", C.c_int32), ("max_newton", C.c_int32), ("barrier_shrink", C.c_double)]


METHODS = {"auto": 0, "lbfgs": 1, "newton": 2}


class Stats(C.Structure):
    _fields_ = [("evals", C.c_int32), ("iters", C.c_int32), ("status", C.c_int32), ("n_ranks", C.c_int32),
                ("dual_value", C.c_double), ("primal_value", C.c_double), ("gap", C.c_double),
                ("infeas", C.c_double), ("wall_seconds", C.c_double), ("device_seconds", C.c_double),
                ("pg", C.c_double), ("pool_subproblems", C.c_int64),
                ("barrier_mu", C.c_double), ("newton_steps", C.c_int32), ("method", C.c_int32)]

    def asdict(self):
        # (one struct.unpack of the record instead of a getattr per field: 6 -> 1 us of the ~500 us of a C3 solve)
        return dict(zip(_STATS_NAMES, _STATS_STRUCT.unpack(bytes(self))))


_STATS_NAMES = tuple(k for k, _ in Stats._fields_)
_STATS_STRUCT = struct.Struct("@" + "".join({C.c_int32: "i", C.c_double: "d", C.c_int64: "q"}[t] for _, t in Stats._fields_))
assert _STATS_STRUCT.size == C.sizeof(Stats)


def build(force=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in ("cfmm_hip.hip", "kernels.hpp", "iterate.hpp", "tiny.hpp", "onewave.hpp", "reorder.hpp", "oneshot.hpp", "pool_math.hpp", "phi2.hpp", "lbfgs_rules.hpp", "smooth.hpp", "chol.hpp", "chol2.hpp", "phik.hpp", "handoff.hpp")]
    srcs.append(os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "cfmm.h"))
    if force or not os.path.exists(_SO) or any(os.path.getmtime(_SO) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _CSRC, "-s"])
    return _SO


_lib = None

SYMBOLS = ["cfmm_create", "cfmm_clone", "cfmm_destroy", "cfmm_last_error", "cfmm_backend", "cfmm_default_opts",
           "cfmm_upload_pools2", "cfmm_upload_poolsN", "cfmm_upload_poolsG", "cfmm_set_pool_flags", "cfmm_set_pool_flagsG", "cfmm_set_utility",
           "cfmm_set_ties", "cfmm_set_deterministic", "cfmm_debug_eval_limbs", "cfmm_eval_dual", "cfmm_eval_smooth", "cfmm_debug_cholesky", "cfmm_debug_cholesky_apply", "cfmm_time_xcd_handoff", "cfmm_solve", "cfmm_solve_batch", "cfmm_batch_capacity", "cfmm_solve_sweep", "cfmm_get_nu", "cfmm_set_nu", "cfmm_get_psi",
           "cfmm_get_solution", "cfmm_get_trades2", "cfmm_get_tradesN", "cfmm_get_tradesG", "cfmm_comm_unique_id", "cfmm_comm_init",
           "cfmm_oneshot_export", "cfmm_oneshot_import", "cfmm_oneshot_attach", "cfmm_oneshot_mailbox", "cfmm_oneshot_enable",
           "cfmm_time_eval_kernel", "cfmm_time_collective", "cfmm_time_newton_kernels", "cfmm_selftest", "cfmm_debug_timers", "cfmm_clock_probe_start", "cfmm_clock_probe_read", "cfmm_clock_probe_stop", "cfmm_clock_probe_chain", "cfmm_pool_count", "cfmm_eval_bytes", "cfmm_stream"]


def lib():
    """Load libcfmm_hip.so (does not need a GPU; creating a context does)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise CfmmError(f"{_SO} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        f"or `make -C {_CSRC}`; there is no CPU fallback")
    L = C.CDLL(_SO)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p
    L.cfmm_create.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
    L.cfmm_clone.argtypes = [vp, C.POINTER(vp)]
    L.cfmm_destroy.argtypes = [vp]
    L.cfmm_last_error.restype = C.c_char_p; L.cfmm_last_error.argtypes = [vp]
    L.cfmm_backend.restype = C.c_char_p; L.cfmm_backend.argtypes = [vp]
    L.cfmm_default_opts.restype = None; L.cfmm_default_opts.argtypes = [C.POINTER(Opts)]
    L.cfmm_upload_pools2.argtypes = [vp, C.c_int, C.c_int64, dp, dp, dp, dp, ip, ip]
    L.cfmm_upload_poolsN.argtypes = [vp, C.c_int, C.c_int64, ip, dp, dp, dp]
    L.cfmm_upload_poolsG.argtypes = [vp, C.c_int, C.c_int, C.c_int64, ip, dp, dp, dp]
    L.cfmm_set_pool_flags.argtypes = [vp, C.c_int, ip]
    L.cfmm_set_pool_flagsG.argtypes = [vp, C.c_int, ip]
    L.cfmm_set_utility.argtypes = [vp, dp, dp, ip]
    L.cfmm_set_ties.argtypes = [vp, C.c_int, ip, dp]
    L.cfmm_set_deterministic.argtypes = [vp, C.c_int]
    L.cfmm_debug_eval_limbs.argtypes = [vp, dp, C.c_double, C.c_double, C.POINTER(C.c_uint64)]
    L.cfmm_eval_dual.argtypes = [vp, dp, dp, dp, dp]
    L.cfmm_eval_smooth.argtypes = [vp, dp, C.c_double, dp, dp, dp, dp]
    L.cfmm_debug_cholesky.argtypes = [vp, C.c_int, dp, dp, dp, ip]
    L.cfmm_debug_cholesky_apply.argtypes = [vp, C.c_int, dp, dp]
    L.cfmm_time_xcd_handoff.argtypes = [vp, C.c_int, C.c_int, dp]
    L.cfmm_solve.argtypes = [vp, dp, C.POINTER(Opts), C.POINTER(Stats)]
    L.cfmm_solve_batch.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(dp), C.POINTER(Opts), C.POINTER(Stats)]
    L.cfmm_batch_capacity.argtypes = [C.c_int]
    L.cfmm_solve_sweep.argtypes = [vp, C.c_int, dp, dp, ip, dp, C.c_int64, ip, ip, dp, dp, dp, C.POINTER(Opts), C.c_double, C.c_int,
                                   dp, dp, dp, ip, dp, C.POINTER(Stats), ip]
    L.cfmm_get_nu.argtypes = [vp, dp]; L.cfmm_set_nu.argtypes = [vp, dp]; L.cfmm_get_psi.argtypes = [vp, dp]
    L.cfmm_get_solution.argtypes = [vp, dp, dp]
    L.cfmm_get_trades2.argtypes = [vp, C.c_int, dp, dp]
    L.cfmm_get_tradesN.argtypes = [vp, C.c_int, dp, dp]
    L.cfmm_get_tradesG.argtypes = [vp, C.c_int, C.c_int, dp, dp]
    L.cfmm_comm_unique_id.argtypes = [C.c_void_p]
    L.cfmm_comm_init.argtypes = [vp, C.c_int, C.c_int, C.c_void_p]
    L.cfmm_oneshot_export.argtypes = [vp, C.c_void_p]
    L.cfmm_oneshot_import.argtypes = [vp, C.c_int, C.c_int, C.c_void_p]
    L.cfmm_oneshot_attach.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.cfmm_oneshot_mailbox.restype = vp; L.cfmm_oneshot_mailbox.argtypes = [vp]
    L.cfmm_oneshot_enable.argtypes = [vp, C.c_int]
    L.cfmm_time_eval_kernel.argtypes = [vp, C.c_int, C.c_int, dp]
    L.cfmm_time_collective.argtypes = [vp, C.c_int, dp, dp]
    L.cfmm_time_newton_kernels.argtypes = [vp, C.c_double, C.c_int, dp]
    L.cfmm_selftest.argtypes = [vp]
    L.cfmm_debug_timers.argtypes = [vp, C.POINTER(C.c_int64)]
    L.cfmm_clock_probe_start.argtypes = [vp, C.c_double, C.c_double]
    L.cfmm_clock_probe_read.argtypes = [vp, C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int)]
    L.cfmm_clock_probe_stop.argtypes = [vp, C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int)]
    L.cfmm_clock_probe_chain.argtypes = [vp, C.POINTER(C.c_int64)]
    L.cfmm_pool_count.restype = C.c_int64; L.cfmm_pool_count.argtypes = [vp]
    L.cfmm_eval_bytes.restype = C.c_int64; L.cfmm_eval_bytes.argtypes = [vp]
    L.cfmm_stream.restype = vp; L.cfmm_stream.argtypes = [vp]
    _lib = L
    return L


_D0 = C.c_double * 0


def _d(a):
    """pointer argument for a float64 array.  A zero-length ctypes array over the buffer converts to POINTER(c_double) like
    data_as does, keeps the array alive, and costs 0.6 us instead of 2.1 (three of them per solve); read-only arrays take the
    slow way"""
    if a is None:
        return None
    try:
        return _D0.from_buffer(a)
    except (TypeError, ValueError):
        return a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


class Context:
    """Thin object wrapper over one cfmm_ctx (one GPU)."""
    second_order = True        # cfmm_solve offers CFMM_METHOD_NEWTON

    def __init__(self, n_tokens, device=0, _handle=None):
        self.L = lib()
        self.n = int(n_tokens)
        if _handle is not None:
            self.h = _handle
            return
        h = C.c_void_p()
        rc = self.L.cfmm_create(int(device), self.n, C.byref(h))
        if rc != 0:
            raise CfmmError(f"cfmm_create failed ({rc}): {self.L.cfmm_last_error(None).decode()}")
        self.h = h

    def clone(self):
        """a context sharing this one's resident pools, with its own stream / utility / solver state"""
        h = C.c_void_p()
        self._chk(self.L.cfmm_clone(self.h, C.byref(h)))
        return Context(self.n, _handle=h)

    def close(self):
        if getattr(self, "h", None):
            self.L.cfmm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise CfmmError(f"libcfmm_hip error {rc}: {self.L.cfmm_last_error(self.h).decode()}", code=int(rc))

    @property
    def backend(self):
        return self.L.cfmm_backend(self.h).decode()

    def upload_pools2(self, kind, Ra, Rb, fee, ia, ib, param=None):
        Ra, Rb, fee, param, ia, ib = f64(Ra), f64(Rb), f64(fee), f64(param), i32(ia), i32(ib)
        self._chk(self.L.cfmm_upload_pools2(self.h, kind, len(Ra), _d(Ra), _d(Rb), _d(fee), _d(param), _i(ia), _i(ib)))

    def upload_poolsN(self, idx, R, w, fee):
        idx, R, w, fee = i32(idx), f64(R), f64(w), f64(fee)
        k, m = R.shape
        self._chk(self.L.cfmm_upload_poolsN(self.h, k, m, _i(idx), _d(R), _d(w), _d(fee)))

    def upload_poolsG(self, kind, idx, R, fee, param=None):
        """a bucket of the K-asset table (POOLK_*: csrc/phik.hpp): idx, R [k][m], fee [m], param [m] or None"""
        idx, R, fee = i32(idx), f64(R), f64(fee)
        param = None if param is None else f64(param)
        k, m = R.shape
        self._chk(self.L.cfmm_upload_poolsG(self.h, kind, k, m, _i(idx), _d(R), _d(fee), _d(param) if param is not None else None))

    def set_pool_flags(self, kind, flags):
        flags = i32(flags)
        self._chk(self.L.cfmm_set_pool_flags(self.h, kind, _i(flags)))

    def set_pool_flagsG(self, k, flags):
        """per-leg tie flags [k][m] of the K-asset table's constant-sum bucket of k tokens (None: none)"""
        flags = i32(flags)
        self._chk(self.L.cfmm_set_pool_flagsG(self.h, int(k), _i(flags)))

    def set_utility(self, c, h=None, ctype=None):
        c, h, ctype = f64(c), f64(h), i32(ctype)
        self._chk(self.L.cfmm_set_utility(self.h, _d(c), _d(h), _i(ctype)))

    def set_ties(self, grp=None, off=None):
        if grp is None:
            self._chk(self.L.cfmm_set_ties(self.h, 0, None, None))
        else:
            grp, off = i32(grp), f64(off)
            self._chk(self.L.cfmm_set_ties(self.h, int(grp.max()) + 1, _i(grp), _d(off)))

    def set_deterministic(self, on=True):
        """bitwise-reproducible psi (integer accumulation): see include/cfmm.h"""
        self._chk(self.L.cfmm_set_deterministic(self.h, 1 if on else 0))

    def debug_eval_limbs(self, nu, ref_reserve, ref_fee):
        """raw fixed-point limbs [3][n] of psi (uint64) from one reproducible-mode evaluation (test hook)"""
        nu = f64(nu)
        limbs = np.zeros((3, self.n), dtype=np.uint64)
        self._chk(self.L.cfmm_debug_eval_limbs(self.h, _d(nu), float(ref_reserve), float(ref_fee), limbs.ctypes.data_as(C.POINTER(C.c_uint64))))
        return limbs

    def eval_dual(self, nu, want_diag=False):
        nu = f64(nu)
        psi = np.zeros(self.n)
        diag = np.zeros(self.n) if want_diag else None
        arb = C.c_double()
        self._chk(self.L.cfmm_eval_dual(self.h, _d(nu), C.byref(arb), _d(psi), _d(diag)))
        return (arb.value, psi, diag) if want_diag else (arb.value, psi)

    def eval_smooth(self, nu, mu, want_hessian=False):
        """barrier-smoothed evaluation (two-asset pools): value, nu'(L - D), psi_mu[, lower-triangular H]"""
        nu = f64(nu)
        psi = np.zeros(self.n)
        H = np.zeros((self.n, self.n), order="F") if want_hessian else None
        val, tr = C.c_double(), C.c_double()
        self._chk(self.L.cfmm_eval_smooth(self.h, _d(nu), float(mu), C.byref(val), C.byref(tr), _d(psi), _d(H)))
        return (val.value, tr.value, psi, H) if want_hessian else (val.value, tr.value, psi)

    def debug_cholesky(self, A, b):
        """solve A x = b with the library's dense Cholesky (A: n x n SPD, n = this context's token count)"""
        A = np.asfortranarray(A, dtype=np.float64); b = f64(b)
        x = np.zeros(self.n); info = C.c_int32()
        self._chk(self.L.cfmm_debug_cholesky(self.h, self.n, _d(A), _d(b), _d(x), C.byref(info)))
        return x, info.value

    def debug_cholesky_apply(self, b):
        """A^-1 b for a new right-hand side through the factor the last debug_cholesky left (the chord step's two products)"""
        b = f64(b); x = np.zeros(self.n)
        self._chk(self.L.cfmm_debug_cholesky_apply(self.h, self.n, _d(b), _d(x)))
        return x

    def time_xcd_handoff(self, np_doubles=2048, reps=50):
        """cfmm_time_xcd_handoff: dict(handoff_us_median, handoff_us_max, flag_seen_us_median, wrong_xcc, failed, orphan_xcds)"""
        out = np.zeros(7)
        self._chk(self.L.cfmm_time_xcd_handoff(self.h, int(np_doubles), int(reps), _d(out)))
        return dict(handoff_us_median=float(out[0]), handoff_us_max=float(out[1]), flag_seen_us_median=float(out[2]),
                    wrong_xcc=int(out[3]), failed=int(out[4]), orphan_xcds=int(out[5]), xcc_of_workgroups_0_to_7="%08d" % int(out[6]))

    def default_opts(self):
        o = Opts()
        self.L.cfmm_default_opts(C.byref(o))
        return o

    def _opts(self, kw):
        d = getattr(self, "_opts0", None)
        if d is None:                       # the defaults are fetched once per context (a ctypes call each is 5 % of a 0.7 ms solve)
            d = self._opts0 = bytes(self.default_opts())
        key = tuple(kw.items())
        memo = getattr(self, "_opts_memo", None)
        if memo is not None and memo[0] == key:           # (the library takes the options as const: the same record serves every solve with the same arguments)
            return memo[1]
        o = Opts.from_buffer_copy(d)
        kw = dict(kw)
        if "tol" in kw:
            o.tol_gap = o.tol_infeas = kw.pop("tol")
        if isinstance(kw.get("method"), str):
            kw["method"] = METHODS[kw["method"]]
        for k, v in kw.items():
            if not hasattr(o, k):
                raise TypeError(f"unknown solver option {k!r}")
            setattr(o, k, v)
        self._opts_memo = (key, o)
        return o

    def batch_capacity(self):
        """solves that cfmm_solve_batch takes per call at this token count"""
        return int(self.L.cfmm_batch_capacity(self.n))

    def solve_batch(self, clones, nu0s=None, **kw):
        """this context and `clones` (its clone()s), each with its own utility set, solved in lock-step: every pool
        column is read once per outer iteration for all of them (cfmm_solve_batch).  Returns one stats dict per solve."""
        ctxs = [self] + list(clones)
        nb = len(ctxs)
        o = self._opts(kw)
        hs = (C.c_void_p * nb)(*[c.h for c in ctxs])
        keep = [None if nu0s is None or nu0s[b] is None else f64(nu0s[b]) for b in range(nb)]
        ptrs = (C.POINTER(C.c_double) * nb)(*[_d(a) if a is not None else C.POINTER(C.c_double)() for a in keep])
        st = (Stats * nb)()
        self._chk(self.L.cfmm_solve_batch(hs, nb, ptrs, C.byref(o), st))
        return [st[b].asdict() for b in range(nb)]

    def solve_sweep(self, c, h, ctype, nu0, sum2=None, trades_per_point=0, kink_tol=1e-3, max_rounds=6, **kw):
        """B utilities over this context's (tiny) network in lock-step, the constant-sum kink loop included (cfmm_solve_sweep:
        two-asset.py:34-100 as one call).  c, h, ctype, nu0: [B][n]; sum2: the constant-sum bucket's columns (dict ia, ib, fee, Ra, Rb) or
        None; trades_per_point: doubles of tenders per point (0: none).  Returns nu, psi [B][n], theta [B][m_sum], tsgn, trades
        [B][T] or None, a list of stats dicts, rounds [B]."""
        c = np.ascontiguousarray(c, dtype=np.float64)
        B, n = c.shape
        h = None if h is None else np.ascontiguousarray(h, dtype=np.float64)
        ctype = None if ctype is None else np.ascontiguousarray(ctype, dtype=np.int32)
        nu0 = np.ascontiguousarray(nu0, dtype=np.float64)
        o = self._opts(kw)
        m = 0 if sum2 is None else len(sum2["Ra"])
        cols = [None] * 5
        if m:
            cols = [i32(sum2["ia"]), i32(sum2["ib"]), f64(sum2["fee"]), f64(sum2["Ra"]), f64(sum2["Rb"])]
        nu = np.empty((B, n)); psi = np.empty((B, n))
        theta = np.empty((B, m)); tsgn = np.zeros((B, m), dtype=np.int32)
        trades = np.empty((B, trades_per_point)) if trades_per_point else None
        st = (Stats * B)()
        rounds = np.zeros(B, dtype=np.int32)
        self._chk(self.L.cfmm_solve_sweep(self.h, B, _d(c), _d(h), _i(ctype), _d(nu0), m, _i(cols[0]), _i(cols[1]), _d(cols[2]), _d(cols[3]), _d(cols[4]),
                                          C.byref(o), float(kink_tol), int(max_rounds), _d(nu), _d(psi), _d(theta) if m else None, _i(tsgn) if m else None,
                                          _d(trades) if trades is not None else None, st, _i(rounds)))
        return nu, psi, theta, tsgn, trades, [st[b].asdict() for b in range(B)], rounds

    def solve(self, nu0=None, **kw):
        o = self._opts(kw)
        st = Stats()
        nu0 = f64(nu0)
        self._chk(self.L.cfmm_solve(self.h, _d(nu0), C.byref(o), C.byref(st)))
        return st.asdict()

    def get_nu(self):
        a = np.zeros(self.n); self._chk(self.L.cfmm_get_nu(self.h, _d(a))); return a

    def set_nu(self, nu):
        nu = f64(nu); self._chk(self.L.cfmm_set_nu(self.h, _d(nu)))

    def get_psi(self):
        a = np.zeros(self.n); self._chk(self.L.cfmm_get_psi(self.h, _d(a))); return a

    def get_solution(self):
        nu = np.zeros(self.n); psi = np.zeros(self.n)
        self._chk(self.L.cfmm_get_solution(self.h, _d(nu), _d(psi)))
        return nu, psi

    def get_trades2(self, kind, m):
        d = np.zeros((2, m)); l = np.zeros((2, m))
        if m:
            self._chk(self.L.cfmm_get_trades2(self.h, kind, _d(d), _d(l)))
        return d, l

    def get_tradesN(self, k, m):
        d = np.zeros((k, m)); l = np.zeros((k, m))
        if m:
            self._chk(self.L.cfmm_get_tradesN(self.h, k, _d(d), _d(l)))
        return d, l

    def get_tradesG(self, kind, k, m):
        d = np.zeros((k, m)); l = np.zeros((k, m))
        if m:
            self._chk(self.L.cfmm_get_tradesG(self.h, kind, k, _d(d), _d(l)))
        return d, l

    def comm_init(self, n_ranks, rank, uid):
        buf = C.create_string_buffer(bytes(uid), 128)
        self._chk(self.L.cfmm_comm_init(self.h, n_ranks, rank, buf))

    def oneshot_export(self):
        """this rank's mailbox of the one-shot xGMI all-reduce as a 64-byte IPC handle (csrc/oneshot.hpp)"""
        buf = C.create_string_buffer(64)
        self._chk(self.L.cfmm_oneshot_export(self.h, buf))
        return bytes(buf.raw)

    def oneshot_import(self, n_ranks, rank, handles):
        buf = C.create_string_buffer(b"".join(bytes(h) for h in handles), 64 * n_ranks)
        self._chk(self.L.cfmm_oneshot_import(self.h, n_ranks, rank, buf))

    def oneshot_enable(self, on):
        self._chk(self.L.cfmm_oneshot_enable(self.h, 1 if on else 0))

    def oneshot_mailbox(self):
        return self.L.cfmm_oneshot_mailbox(self.h)

    def oneshot_attach(self, n_ranks, rank, mailboxes):
        """same-process ranks (tests): raw device pointers of every rank's mailbox"""
        arr = (C.c_void_p * n_ranks)(*[C.c_void_p(m) for m in mailboxes])
        self._chk(self.L.cfmm_oneshot_attach(self.h, n_ranks, rank, arr))

    def time_eval_kernel(self, kind, reps=20):
        s = C.c_double()
        self._chk(self.L.cfmm_time_eval_kernel(self.h, kind, reps, C.byref(s)))
        return s.value

    def time_collective(self, reps=50):
        """(fold seconds, all-reduce seconds) per launch; collective when a communicator is set"""
        f, a = C.c_double(), C.c_double()
        self._chk(self.L.cfmm_time_collective(self.h, reps, C.byref(f), C.byref(a)))
        return f.value, a.value

    def time_newton_kernels(self, mu, reps=5):
        """seconds per launch group of one second-order step at the current prices: smoothed evaluation with the Hessian,
        smoothed evaluation alone, dense factorisation, back substitution"""
        out = np.zeros(4)
        self._chk(self.L.cfmm_time_newton_kernels(self.h, float(mu), int(reps), _d(out)))
        return dict(smooth_hess=out[0], smooth=out[1], factor=out[2], backsolve=out[3])

    def selftest(self):
        self._chk(self.L.cfmm_selftest(self.h))

    def debug_timers(self):
        a = np.zeros(64 + 8 * 4096 + 2048, dtype=np.int64)
        self._chk(self.L.cfmm_debug_timers(self.h, a.ctypes.data_as(C.POINTER(C.c_int64))))
        return a[:64].reshape(32, 2), a[64:64 + 8 * 4096].reshape(4096, 8), a[64 + 8 * 4096:].reshape(1024, 2)

    def clock_probe_start(self, period_us=100.0, max_ms=2000.0):
        """one sleeping wave on a stream of its own samples {shader cycles, 100 MHz ticks} every period_us while the caller's work runs"""
        self._chk(self.L.cfmm_clock_probe_start(self.h, float(period_us), float(max_ms)))

    def _probe_samples(self, f):
        a = np.zeros((8192, 2), dtype=np.int64)
        k = C.c_int()
        self._chk(f(self.h, a.ctypes.data_as(C.POINTER(C.c_int64)), a.shape[0], C.byref(k)))
        return a[:k.value].copy()

    def clock_probe_read(self):
        """samples so far, [k, 2] int64 (shader cycles, 100 MHz ticks); no synchronisation"""
        return self._probe_samples(self.L.cfmm_clock_probe_read)

    def clock_probe_stop(self):
        return self._probe_samples(self.L.cfmm_clock_probe_stop)

    def clock_probe_chain(self):
        """(shader cycles, 100 MHz ticks, links) of the probe's dependent-FMA chain"""
        a = np.zeros(3, dtype=np.int64)
        self._chk(self.L.cfmm_clock_probe_chain(self.h, a.ctypes.data_as(C.POINTER(C.c_int64))))
        return int(a[0]), int(a[1]), int(a[2])

    def pool_count(self):
        return int(self.L.cfmm_pool_count(self.h))

    def eval_bytes(self):
        """bytes of pool columns one dual evaluation loads as stored now (compact mirrors where built)"""
        return int(self.L.cfmm_eval_bytes(self.h))


def comm_unique_id():
    buf = C.create_string_buffer(128)
    rc = lib().cfmm_comm_unique_id(buf)
    if rc != 0:
        raise CfmmError(f"cfmm_comm_unique_id failed ({rc}): {lib().cfmm_last_error(None).decode()}")
    return bytes(buf.raw)
