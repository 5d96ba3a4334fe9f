"""CPU suite, part 2: host logic of the product (packing, start prices, kink ties + fill recovery,
sharding) with the C oracle standing in for the device (tests/oracle_ctx.py), and the C-ABI
library itself: it must load and export every symbol include/cfmm.h declares, and must refuse to
run without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import cfmm
from cfmm import _lib, synthetic
from oracle import instances as I
from oracle.primal_scipy import solve_primal
from helpers import golden, shipped_cases, problem_of, random_instance, normalise_with_params
from oracle_ctx import OracleContext

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_exports_header_symbols():
    """two-sided: every entry point include/cfmm.h declares is exported, and the library exports no cfmm_* symbol
    the header does not declare (tuning-build-only entry points sit behind #ifdef in both places)"""
    import subprocess
    _lib.build()
    L = ctypes.CDLL(_lib._SO)
    header = open(os.path.join(ROOT, "include", "cfmm.h")).read()
    header = re.sub(r"#ifdef CFMM_SMOOTH_HIST.*?#endif", "", header, flags=re.S)       # (not part of the shipped build)
    names = set(re.findall(r"\b(cfmm_[a-z_0-9A-Z]+)\s*\(", header))
    assert len(names) >= 20
    for nm in names:
        assert hasattr(L, nm), f"libcfmm_hip.so lacks {nm}"
    assert names == set(_lib.SYMBOLS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib._SO]).decode()
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("cfmm_")}
    assert exported == names, f"undeclared exports: {sorted(exported - names)}; missing: {sorted(names - exported)}"


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """cfmm_opts / cfmm_stats as bound by cfmm/_lib.py AND by the verbatim stub in INTEGRATION.md have the size and
    field offsets the C compiler gives the header's structs (a mismatch corrupts memory silently)"""
    import subprocess
    src = tmp_path / "layout.c"
    fields_o = [f for f, _ in _lib.Opts._fields_]
    fields_s = [f for f, _ in _lib.Stats._fields_]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "cfmm.h"', 'int main(void) {',
             'printf("%zu %zu\\n", sizeof(cfmm_opts), sizeof(cfmm_stats));']
    lines += [f'printf("%zu\\n", offsetof(cfmm_opts, {f}));' for f in fields_o]
    lines += [f'printf("%zu\\n", offsetof(cfmm_stats, {f}));' for f in fields_s]
    lines += ['return 0; }']
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert int(out[0]) == ctypes.sizeof(_lib.Opts) and int(out[1]) == ctypes.sizeof(_lib.Stats)
    offs = [int(x) for x in out[2:]]
    assert offs[:len(fields_o)] == [getattr(_lib.Opts, f).offset for f in fields_o]
    assert offs[len(fields_o):] == [getattr(_lib.Stats, f).offset for f in fields_s]
    # the stub in INTEGRATION.md declares the same two structs by hand
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = md[md.index("import ctypes as C, numpy as np"):]
    stub = stub[:stub.index("```")]
    decl = stub[stub.index("class Opts"):stub.index("L.cfmm_last_error.restype")]
    ns = {"C": ctypes}
    exec(decl, ns)
    assert [f for f, _ in ns["Opts"]._fields_] == fields_o and ctypes.sizeof(ns["Opts"]) == ctypes.sizeof(_lib.Opts)
    assert [f for f, _ in ns["Stats"]._fields_] == fields_s and ctypes.sizeof(ns["Stats"]) == ctypes.sizeof(_lib.Stats)


def test_ctypes_plumbing_shortcuts_agree_with_the_plain_forms():
    """the per-solve shortcuts of cfmm/_lib.py (round 4: 34 -> ~20 us of Python per solve): the statistics record unpacked in one
    struct.unpack equals the field-by-field read; a float64 array passed as a zero-length ctypes array over its buffer is the
    same address as ndarray.ctypes.data_as gives (read-only arrays take the slow way); the options record is rebuilt when an
    argument changes and only then"""
    st = _lib.Stats()
    vals = dict(evals=22, iters=21, status=1, n_ranks=1, dual_value=1.5, primal_value=1.25, gap=1e-7, infeas=2e-7, wall_seconds=5e-4,
                device_seconds=4.5e-4, pg=0.125, pool_subproblems=22_000_000, barrier_mu=1e-9, newton_steps=8, method=2)
    for k, v in vals.items():
        setattr(st, k, v)
    assert st.asdict() == {k: getattr(st, k) for k, _ in _lib.Stats._fields_} == vals
    dp = ctypes.POINTER(ctypes.c_double)
    a = np.arange(7.0)
    for arr in (a, a[:0].copy(), np.zeros((3, 4))):
        assert ctypes.addressof(_lib._d(arr)) == arr.ctypes.data or arr.size == 0
    ro = np.arange(5.0); ro.flags.writeable = False
    assert ctypes.cast(_lib._d(ro), ctypes.c_void_p).value == ro.ctypes.data and isinstance(_lib._d(ro), dp)
    assert _lib._d(None) is None

    class Ctx(_lib.Context):                  # (the options logic alone: no library behind it)
        def __init__(self):
            self.calls = 0
        def default_opts(self):
            self.calls += 1
            o = _lib.Opts(); o.tol_gap = o.tol_infeas = 1e-6; o.max_evals = 2000
            return o
        def __del__(self):
            pass
    c = Ctx()
    o1 = c._opts(dict(tol=1e-6, max_evals=100, method="lbfgs"))
    assert c._opts(dict(tol=1e-6, max_evals=100, method="lbfgs")) is o1 and c.calls == 1
    o2 = c._opts(dict(tol=1e-8, max_evals=100, method="lbfgs"))
    assert o2 is not o1 and o2.tol_gap == 1e-8 and o1.tol_gap == 1e-6 and o2.max_evals == 100 and o2.method == _lib.METHODS["lbfgs"]
    with pytest.raises(TypeError):
        c._opts(dict(no_such_option=1))


def test_no_kernel_of_the_library_carries_a_private_segment(tmp_path):
    """`table_newton_kernel<true>` had picked up a 16-byte private segment (three spilled dwords) behind the builder's
    last resource check -- a tool, not a test.  Here the check reads the code object the built library CARRIES (no recompilation): the
    .hip_fatbin section -> the gfx950 bundle -> its AMDGPU metadata note; every kernel must report .private_segment_fixed_size 0, and
    the register-bound kernels must stay inside the budget their launch geometry assumes."""
    import shutil, subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("the ROCm LLVM binutils are not installed")
    so = _lib.build()
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "co.elf")
    subprocess.check_call([tools[0], f"--dump-section=.hip_fatbin={fat}", so, str(tmp_path / "rest.so")])
    subprocess.check_call([tools[1], "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
    notes = subprocess.run([tools[2], "--notes", co], capture_output=True, text=True, check=True).stdout
    rows = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", notes, re.S):
        g = lambda k: int((re.search(r"\.%s:\s+(\d+)" % k, m.group(2)) or [0, 0])[1])
        rows[m.group(1)] = dict(scratch=g("private_segment_fixed_size"), vgpr=g("vgpr_count"), agpr=g("agpr_count"))
    assert len(rows) >= 140, len(rows)                 # (every device kernel of the library: 145 at the time of writing)
    spilled = {k: v for k, v in rows.items() if v["scratch"]}
    assert not spilled, spilled
    for k, v in rows.items():
        if "iter_kernel" in k or ("eval_kernel" in k and "table" not in k) or "eval_batch_kernel" in k:
            assert v["vgpr"] + v["agpr"] <= 128, (k, v)     # 16 waves of 1024 threads per CU: 4 per SIMD
        if "table_eval_kernel" in k or "table_newton_kernel" in k:
            assert v["vgpr"] + v["agpr"] <= 256, (k, v)     # 8 waves of 512 threads: 2 per SIMD


def test_product_fails_loudly_without_gpu():
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is visible")
    except ImportError:
        pass
    with pytest.raises(cfmm.CfmmError, match="no HIP device|no CPU"):
        _lib.Context(4)
    inst = I.arbitrage()
    p = problem_of(inst)
    with pytest.raises(cfmm.CfmmError):
        p.solve()


def test_bench_self_launches_one_rank_per_gpu(tmp_path):
    """`python bench.py --gpus N` (the driver's command, no launcher around it) must start N ranks itself; without
    GPUs every rank then refuses loudly instead of falling back to anything"""
    import subprocess
    import sys
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is visible")
    except ImportError:
        pytest.skip("torch missing")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    # (the launcher stops the surviving rank as soon as one has failed: one or both get to print)
    assert re.search(r"rank [01] of 2: --gpus 2 but only 0 GPU\(s\) are visible", r.stderr), r.stderr[-2000:]


@pytest.mark.parametrize("cfg", ["C2", "C3", "C4", "C5", "C4x4", "C3zipf"])
def test_committed_bench_lines_keep_the_contract(cfg):
    """profiles/r06_bench_*.json: the round's driver-reproducible line of every BASELINE.json configuration that fits one GPU
    (`python bench.py --config Cx` on MI355X, tools/gpu_profile.sh), plus the two lines SURVEY 8(d) asks for beside them (C4x4:
    a pool set that MUST stream from HBM; C3zipf: hub-weighted token pairs) -- BASELINE's metric and unit, whole-job value
    consistent with evaluations x pools / time, BOTH ceilings in the roofline object with `bound` naming the binding one, the
    CPU baseline beside it"""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "r06_bench_%s.json" % cfg)
    d = json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "pool-subproblems/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["config"]["workload"].startswith(("C3" if cfg == "C3zipf" else cfg) + ":")
    assert "model" not in d["config"] and d["scaling"] == ("strong" if cfg in ("C4", "C4x4") else "weak")
    work = d["evals_per_solve"] * d["config"]["pools_total"]
    assert abs(d["ms_per_step"] * 1e-3 * d["value"] - work) <= 1e-6 * work
    assert d["gap"] <= 1e-6 and d["infeas"] <= 1e-6
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "valu") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["avg_launch_us"] > 0
    # round 6: the line explains itself -- blocks of the timed steps, the shader clock measured live (a pass of its own
    # behind the timed region) beside the idle clock, the line's kernel times against profiles/budget.json
    blocks = d["ms_per_step_blocks"]
    assert len(blocks) == 5 and abs(sum(blocks) / 5 - d["ms_per_step"]) <= 0.02 * d["ms_per_step"] and max(blocks) <= 1.12 * min(blocks)
    assert min(blocks) <= d["ms_per_step_median_block"] <= max(blocks) and d["extra_warmup_steps"] >= 100
    assert 1.5 < rf["effective_clock_ghz_live"] <= 2.5 and 2.2 < rf["clock_ghz_idle"] <= 2.5 and rf["clock_pass"]["solves"] >= 20
    assert 5.5 < rf["clock_probe_fma_chain"]["cycles_per_dependent_v_fma_f64"] < 8.0
    if cfg != "C3zipf":
        assert d["budget_ok"] is True and d["budget"]["over"] == {} and d["budget"]["checked"]
    if cfg == "C5":                                   # second-order path: the dense factorisation against the fp64 vector peak
        assert rf["unit"] == "TFLOP/s" and rf["peak"] == 78.6 and rf["bound"] == "valu" and d["newton_steps_per_solve"] >= 3
        assert set(rf["newton_step_us"]) == {"smoothed_evaluation_with_hessian", "smoothed_evaluation", "factorisation", "back_substitution"}
        assert rf["flop_frac"] == rf["frac"] and rf["valu_frac"] is None       # a flop rate, NOT the PMC issue fraction
        # round 5: the kernel the rocprof summary lists, its launch count, the issued and the useful flop fractions
        assert rf["kernel"].startswith("chol_step2_kernel") and rf["launches_per_factorisation"] == 17
        assert abs(rf["useful_flop_frac"] * 3.0 - rf["flop_frac"]) <= 1e-12 and "useful_flop_frac" in rf["flop_frac_note"]
        assert rf["rocprof_source"].endswith("kernel_stats_C5newton.csv") and 10.0 < rf["rocprof_avg_per_launch_us"] < 25.0
        assert abs(rf["rocprof_avg_launch_us"] - rf["avg_launch_us"]) <= 0.12 * rf["avg_launch_us"]     # the chain: trace x launches against HIP events
        assert d["ms_per_step"] <= 4.6                # (round 2: 13.6, round 3: 6.6, round 4: 4.1-4.2)
        sp = d["start_prices"]                        # memoised start prices said out loud, with the un-memoised time beside it
        assert sp["ms_per_step_memoised"] == d["ms_per_step"] and sp["ms_per_step_recomputed"] >= 0.93 * sp["ms_per_step_memoised"]      # (O(n) extra: within the run-to-run noise of two 10-step timings)
        assert "EVALUATIONS only" in d["cpu_baseline"]["measures"]
    else:
        # `frac` prices the algorithmic bytes (the contract), `hbm_frac` the bytes as stored: equal unless a compact mirror exists
        # (round 5: the K-asset buckets' log(R/w) column is counted as stored -- more than the algorithm names where such pools exist)
        assert rf["unit"] == "GB/s" and rf["peak"] == 8000.0
        ratio = rf["bytes_as_stored_per_launch"] / rf["algorithmic_bytes_per_launch"]
        assert abs(rf["hbm_frac"] - rf["frac"] * ratio) <= 1e-9
        mirrored = ratio < 1.0
        assert mirrored == (cfg in ("C4", "C4x4")) and (ratio == 1.0) == (cfg == "C2") and ratio <= 1.15
        assert rf["traffic"] is None or abs(rf["traffic"] - rf["bytes_as_stored_per_launch"]) <= (0.12 if cfg != "C2" else 3.0) * rf["bytes_as_stored_per_launch"]
        assert rf["bound"] == ("valu" if (rf["valu_frac"] or 0.0) > rf["hbm_frac"] else "hbm")
        assert rf["evaluation_only"]["bound"] in ("hbm", "valu") and 0.0 < rf["evaluation_only"]["hbm_frac"] < 1.0
    if cfg != "C3zipf":                               # (the stress variant is timed without the CPU leg)
        cb = d["cpu_baseline"]
        assert cb["kind"] == "port" and cb["unit"] == "pool-subproblems/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
        # BASELINE.md section 4's baseline B beside A: vectorised NumPy on one thread, where the restatement covers the pool kinds and
        # the pool set fits the host's memory several times over
        if cfg in ("C2", "C3", "C4"):
            nb = cb["numpy_one_thread"]
            assert nb["cores"] == 1 and nb["kind"] == "port" and 1e6 < nb["value"] < cb["value"] and "one thread" in nb["sample"]
        else:
            assert "numpy_one_thread" not in cb
    if cfg == "C3":
        assert d["ms_per_step"] <= 0.52 and d["value"] >= 4.2e10      # (round 2: 0.594 ms, 3.70e10; round 5's driver line: 0.574)
        assert rf["avg_launch_us"] <= 20.6 and rf["frac"] >= 0.26     # (rounds 4-5: 20.0-20.9 us)
        # the live clock accounts for the launch: duration x clock in shader cycles, live against the profiled pass
        cc = rf["clock_check"]
        assert 0.9 < cc["time_ratio_live_over_profiled"] < 1.1 and cc["launch_shader_cycles_live"] > 0
    if cfg == "C4":
        assert rf["avg_launch_us"] <= 52.0 and rf["frac"] >= 0.78  # (round 2: 68-69 us, 0.58; round 3: 59.4; before the mirror: 58.1)
        assert rf["bytes_as_stored_per_launch"] == 210_000_000 and rf["hbm_frac"] >= 0.5
        assert "PARTLY CACHE-SERVED" in rf["note"]    # 210 MB as stored against a 256 MiB Infinity Cache is not an HBM figure
    if cfg == "C4x4":                                 # 1.28 GB of algorithmic columns per launch, 0.84 GB as stored: the HBM-streaming figure
        assert rf["algorithmic_bytes_per_launch"] == 1_280_000_000 and rf["bytes_as_stored_per_launch"] == 840_000_000 and rf["bound"] == "hbm"
        assert rf["avg_launch_us"] <= 172.0           # (before the mirror and the wide tiles: 198-208)
        assert rf["hbm_frac"] >= 0.62 and rf["evaluation_only"]["hbm_frac"] >= 0.70 and rf["frac"] >= 0.93
    if cfg == "C3zipf":
        assert "Zipf(1.1)" in d["config"]["workload"] and rf["avg_launch_us"] <= 24.0


def test_product_never_imports_oracle():
    """the oracle is test infrastructure: no import, include, link or dlopen of it in the product"""
    pkg = os.path.join(ROOT, "cfmm-routing-code_amd")
    pat = re.compile(r"^\s*(import\s+oracle|from\s+oracle|#\s*include.*oracle)|dlopen\(.*oracle|libcfmm_oracle|c_oracle", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")) or f == "Makefile":
                src = open(os.path.join(dirpath, f)).read()
                assert not pat.search(src), f"{f} uses the oracle"


def test_pack_roundtrip_and_validation():
    inst = I.arbitrage()
    net, where = cfmm.pack(inst["n_tokens"], inst["local_indices"], inst["reserves"], inst["fees"], inst["kinds"], inst["weights"])
    assert where == [(4, 0), ("cp2", 0), ("cp2", 1), ("cp2", 2), ("sum2", 0)]
    assert net["gn"][4]["idx"].shape == (4, 1) and np.allclose(net["gn"][4]["w"][:, 0], [.4, .3, .2, .1])
    with pytest.raises(ValueError):
        cfmm.pack(3, [[0, 0]], [[1, 1]], [0.99])
    with pytest.raises(ValueError):
        cfmm.pack(3, [[0, 5]], [[1, 1]], [0.99])
    with pytest.raises(ValueError):
        cfmm.pack(3, [[0, 1]], [[1, -1]], [0.99])
    # three-token constant-sum / stableswap pools: the K-asset table's buckets (round 4; they were refused before)
    net3, where3 = cfmm.pack(3, [[0, 1, 2], [2, 0, 1]], [[1, 1, 1], [2, 3, 4]], [0.99, 0.999], kinds=["sum", "curve"], params=[None, 5.0])
    assert where3 == [(("sum", 3), 0), (("stable", 3), 0)] and cfmm.problem.network_pool_count(net3) == 2
    assert net3["gk"][("stable", 3)]["idx"][:, 0].tolist() == [2, 0, 1] and net3["gk"][("stable", 3)]["param"][0] == 5.0
    with pytest.raises(ValueError):
        cfmm.pack(3, [[0, 1, 2]], [[1, 1, 1]], [0.99], kinds=["curve"])              # alpha missing
    with pytest.raises(ValueError):
        cfmm.pack(12, [list(range(9))], [[1.0] * 9], [0.99], kinds=["sum"])          # more than 8 tokens
    # empty problem object
    net, where = cfmm.pack(3, [], [], [])
    assert where == [] and cfmm.problem.network_pool_count(net) == 0


def test_shard_network_partitions_every_bucket():
    net = synthetic.config("C3", scale=0.001)
    parts = [cfmm.shard_network(net, r, 3) for r in range(3)]
    for key in ("cp2", "w2"):
        assert np.array_equal(np.concatenate([p[key]["Ra"] for p in parts]), net[key]["Ra"])
    for k in net["gn"]:
        assert np.array_equal(np.concatenate([p["gn"][k]["R"] for p in parts], axis=1), net["gn"][k]["R"])
    netk = synthetic.config("GK", scale=0.02)
    parts = [cfmm.shard_network(netk, r, 3) for r in range(3)]
    for key in netk["gk"]:
        assert np.array_equal(np.concatenate([p["gk"][key]["R"] for p in parts], axis=1), netk["gk"][key]["R"])
        assert np.array_equal(np.concatenate([p["gk"][key]["param"] for p in parts]), netk["gk"][key]["param"])
    assert sum(cfmm.problem.network_pool_count(p) for p in parts) == cfmm.problem.network_pool_count(netk)


def test_start_prices_propagate_through_pools():
    inst = I.liquidation()
    p = problem_of(inst)
    nu0 = cfmm.start_prices(p.net, p.utility)
    assert nu0[4] == 1.0 and np.all(nu0 > 0) and np.all(np.isfinite(nu0))
    # a consistent network gives back its latent prices exactly
    net = synthetic.make_network(50, m_cp2=400, seed=1, mispricing=0.0)
    c = np.zeros(50); c[7] = net["prices"][7]
    nu0 = cfmm.start_prices(net, cfmm.Utility(c))
    assert np.abs(nu0 / net["prices"] - 1).max() < 1e-9


def test_start_prices_from_network_potentials():
    """cfmm.problem._potentials: one least-squares fit per network (cached in the network dict, not inherited by shards), O(n)
    per utility -- several priced tokens shift it by their mean, a component the utility prices nowhere gets 1, another utility
    on the same network costs no new fit"""
    net = synthetic.make_network(60, m_cp2=600, m_w2=100, m_gn=60, seed=3, mispricing=0.0)
    pi = net["prices"]
    c = np.zeros(60); c[[3, 17, 40]] = pi[[3, 17, 40]] * np.array([1.0, 1.02, 0.98])       # inconsistent by +-2 %: the fit takes their mean
    nu0 = cfmm.start_prices(net, cfmm.Utility(c))
    assert np.array_equal(nu0[[3, 17, 40]], c[[3, 17, 40]])                                  # priced tokens keep their price
    free = np.setdiff1d(np.arange(60), [3, 17, 40])
    shift = np.exp(np.mean(np.log([1.0, 1.02, 0.98])))
    assert np.abs(nu0[free] / (pi[free] * shift) - 1).max() < 1e-9
    fit = net["_potentials"]
    c2 = np.zeros(60); c2[5] = 2.0 * pi[5]
    nu1 = cfmm.start_prices(net, cfmm.Utility(c2))
    assert net["_potentials"] is fit and np.abs(nu1 / (2.0 * pi) - 1).max() < 1e-9          # the same fit, another anchor
    assert "_potentials" not in cfmm.shard_network(net, 0, 2) and "_price_relations" not in cfmm.shard_network(net, 0, 2)
    # two components: tokens 0-29 and 30-59 never share a pool; the utility prices only the first
    a = synthetic.make_network(30, m_cp2=300, seed=4, mispricing=0.0)
    two = dict(n_tokens=60, prices=np.concatenate([a["prices"], a["prices"]]), c=np.concatenate([a["c"], a["c"]]), seed=0,
               cp2={k: np.concatenate([a["cp2"][k], a["cp2"][k] + (30 if k in ("ia", "ib") else 0)]) for k in a["cp2"]})
    c3 = np.zeros(60); c3[7] = a["prices"][7]
    nu2 = cfmm.start_prices(two, cfmm.Utility(c3))
    assert np.abs(nu2[:30] / a["prices"] - 1).max() < 1e-9 and np.all(nu2[30:] == 1.0)


@pytest.mark.parametrize("name,inst", shipped_cases())
def test_host_logic_reproduces_shipped_instances(oracle_lib, name, inst):
    """objective, psi and every pool's tenders, incl. the partially filled constant-sum pool"""
    g = golden()[name]
    p = problem_of(inst, OracleContext(inst["n_tokens"]))
    v = p.solve(tol=1e-10)
    assert p.status == "optimal"
    assert abs(v - g["survey"]["value"]) <= 1e-8 * max(1, abs(v))
    assert abs(v - g["kkt"]["value"]) <= 1e-9 * max(1, abs(v))
    assert p.gap <= 1e-10 and p.infeas <= 1e-10
    # psi and every pool's tenders against the 50-digit KKT solution: 1e-6 absolute is the bar,
    # 2e-8 is what a solve to 1e-10 certificates reaches
    assert np.abs(p.psi - np.asarray(g["kkt"]["psi"])).max() <= 2e-8
    for d, l, y in zip(p.deltas, p.lambdas, g["kkt"]["y"]):
        assert np.all(d >= 0) and np.all(l >= 0) and np.all(d * l == 0)
        assert np.abs((l - d) - np.asarray(y)).max() <= 1e-6
        assert np.abs((l - d) - np.asarray(y)).max() <= 2e-8
    if "nu" in g["survey"] and not name.startswith("two_asset"):   # (token 1's price is not unique there)
        assert np.abs(p.nu / np.asarray(g["survey"]["nu"]) - 1).max() <= 1e-5


def test_two_asset_full_sweep_monotone(oracle_lib):
    """two-asset.py:40-100: u(t) over the 50-point sweep is increasing and concave"""
    vals = []
    nu = None
    for t in I.two_asset_sweep():
        p = problem_of(I.two_asset(t), OracleContext(3))
        vals.append(p.solve(tol=1e-9, nu0=nu)); nu = p.nu
        assert p.status == "optimal"
    vals = np.array(vals)
    assert np.all(np.diff(vals) > 0) and np.all(np.diff(vals, 2) < 1e-7)
    assert abs(vals[0] - 6.2330001314) < 1e-7 and abs(vals[-1] - 44.1820204014) < 1e-7


@pytest.mark.parametrize("seed", range(24))
def test_random_instances_with_constant_sum_pools(oracle_lib, seed):
    util = ["arbitrage", "swap", "liquidate"][seed % 3]
    inst = random_instance(100 + seed, n_tokens=5 + seed % 3, n_pools=10 + seed % 5, with_sum=True,
                           with_curve=(seed % 2 == 1), utility=util)
    p = problem_of(inst, OracleContext(inst["n_tokens"]))
    v = p.solve(tol=1e-9)
    r = solve_primal(normalise_with_params(inst))
    if p.status == "infeasible":          # a token to sell that no pool lists: SLSQP fails on it too
        assert not r["success"]
        return
    assert p.status == "optimal" and p.gap <= 1e-8 and p.infeas <= 1e-8
    assert r["value"] <= v + 2e-6 * max(1, abs(v))
    if r["success"]:
        assert abs(v - r["value"]) <= 2e-6 * max(1, abs(v)), (v, r["value"], p._theta)


def test_barrier_smoothed_constant_sum_pool_against_finite_differences(tmp_path):
    """csrc/phik.hpp: sum_smooth_k -- the K-asset constant-sum pool (arbitrage.py:73-74 over K tokens) with the second-order path's log
    barrier on its sign constraints -- is __host__ __device__: compiled for the HOST here and checked without a GPU.  The pool's
    constraint holds to rounding, the value tends to the LP's as the weight shrinks, its gradient is L - D and its Hessian the closed
    form, both against central differences."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc missing")
    src = tmp_path / "t.hip"
    src.write_text(r'''
#include "phik.hpp"
#include <cstdio>
#include <cmath>
using namespace cfmm;
int main()
{
    constexpr int K = 4;
    double R[K] = {31.5, 29.4, 27.2, 15.5}, p[K] = {1.005, 1.007, 1.0, 1.0153};
    const double g = 0.99;
    for (double mu : {1e-1, 1e-3, 1e-6, 1e-10}) {
        SumSmooth<K> o; sum_smooth_k<K>(R, p, g, mu, o);
        double bal = 0, minslack = 1e300;
        for (int j = 0; j < K; ++j) { bal += g * o.D[j] - o.L[j]; minslack = std::fmin(minslack, std::fmin(std::fmin(o.L[j], R[j] - o.L[j]), o.D[j])); }
        double maxg = 0, maxh = 0;
        for (int k = 0; k < K; ++k) {
            const double e = 1e-6 * std::fmax(mu, 1e-4);
            double pp[K], pm[K]; for (int j = 0; j < K; ++j) { pp[j] = pm[j] = p[j]; } pp[k] += e; pm[k] -= e;
            SumSmooth<K> a, b; sum_smooth_k<K>(R, pp, g, mu, a); sum_smooth_k<K>(R, pm, g, mu, b);
            maxg = std::fmax(maxg, std::fabs((a.val - b.val) / (2 * e) - (o.L[k] - o.D[k])));
            for (int j = 0; j < K; ++j) {
                const double fh = ((a.L[j] - a.D[j]) - (b.L[j] - b.D[j])) / (2 * e);
                maxh = std::fmax(maxh, std::fabs(fh - sum_smooth_hess<K>(o, g, j, k)) / (1.0 + std::fabs(fh)));
            }
        }
        printf("%.0e %.12g %.12g %.3e %.3e %.3e %.3e\n", mu, o.val, o.trade, bal, minslack, maxg, maxh);
    }
    return 0;
}
''')
    exe = tmp_path / "t"
    inc = os.path.join(ROOT, "cfmm-routing-code_amd", "csrc")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-w", "-I", inc, "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60).stdout.strip().splitlines()
    rows = [[float(x) for x in l.split()] for l in out]
    assert [r[0] for r in rows] == [1e-1, 1e-3, 1e-6, 1e-10]
    # the LP's value at these prices: token 5 (price 1.0) pays for the whole of token 2 (price 1.0153): R (p - 1 / gamma)
    lp = 15.5 * (1.0153 - 1.0 / 0.99)
    for mu, val, trade, bal, minslack, maxg, maxh in rows:
        assert abs(bal) <= 1e-12 * 100.0 and minslack > 0.0                      # feasible, strictly inside
        assert lp - 1e-12 <= trade + 12 * mu and trade <= lp + 1e-9              # p'y within 3 K mu of the LP's value, never above it
        assert maxg <= 1e-4 and maxh <= 2e-3                                     # (central differences at step 1e-6 max(mu, 1e-4))
    assert abs(rows[-1][2] - lp) <= 1e-8


def test_switch_records_of_a_k_asset_constant_sum_pool_are_rooted_at_the_paying_leg():
    """cfmm/problem.py: _canonical_switches, _split_payers (round 6, pure functions of the records: every rank of a pool-sharded solve
    computes the same).  Three tokens tied for cheapest, found pair by pair in rounds whose cheapest token differed -- records (0, 3) and
    (2, 3) -- become (0, 2) and (0, 3): rooted at the leg the device makes pay (the lowest of the tied legs), same union of tied tokens.
    A partly drained leg of such a pool gets one record per cheapest token that could pay for it."""
    from cfmm.problem import Problem, _drain_leg
    pidx, pR = [10, 11, 12, 13], [5.0, 6.0, 7.0, 8.0]
    sw = lambda a, b: dict(sgn=0, ia=pidx[a], ib=pidx[b], fee=0.99, Ra=0.0, Rb=0.0, loose=False, leg_a=a, leg_b=b, pool=7, k=4, pidx=pidx, pR=pR)
    tied = {(0, 4, 7, 103): sw(0, 3), (0, 4, 7, 123): sw(2, 3)}
    out = Problem._canonical_switches(tied, set())
    assert sorted(out) == [(0, 4, 7, 102), (0, 4, 7, 103)]
    assert [(r["leg_a"], r["leg_b"], r["ia"], r["ib"]) for r in out.values()] == [(0, 2, 10, 12), (0, 3, 10, 13)]
    assert Problem._canonical_switches(out, set()) == out                                   # (a fixed point)
    assert sorted(Problem._canonical_switches(tied, {((0, 4, 7, 102), 0)})) == [(0, 4, 7, 103)]      # (a banned record is not re-created)
    # a drain record of leg 1 paid for by leg 0, with legs 0, 2, 3 tied for cheapest: one more record per other payer
    drain = dict(sgn=1, ia=pidx[0], ib=pidx[1], fee=0.99, Ra=pR[0], Rb=pR[1], loose=False, leg_lo=0, pidx=pidx, pR=pR)
    both = dict(out); both[(0, 4, 7, 1)] = drain
    sp = Problem._split_payers(dict(sorted(both.items())), set())
    assert sorted(k[3] for k in sp) == [1, 102, 103, 221, 231]
    assert (sp[(0, 4, 7, 221)]["ia"], sp[(0, 4, 7, 221)]["leg_lo"], sp[(0, 4, 7, 221)]["Ra"], sp[(0, 4, 7, 221)]["ib"]) == (12, 2, 7.0, 11)
    assert [_drain_leg(c) for c in (1, 221, 231)] == [1, 1, 1]
    assert Problem._split_payers(sp, set()) == sp
    # a pool without tied payers, and a payer that is not among the tied tokens: nothing to split
    assert Problem._split_payers({(0, 4, 7, 1): drain}, set()) == {(0, 4, 7, 1): drain}
    other = dict(drain, leg_lo=1, ia=pidx[1], ib=pidx[2], Ra=pR[1], Rb=pR[2])
    lone = {(0, 4, 7, 2): other, (0, 4, 7, 103): sw(0, 3)}
    assert Problem._split_payers(dict(sorted(lone.items())), set()) == dict(sorted(lone.items()))
