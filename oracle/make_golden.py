"""TEST INFRASTRUCTURE -- regenerates tests/golden/shipped_instances.json.

    python -B oracle/make_golden.py

Three independent sources per instance, all stored:
  * "kkt":    the KKT system of the program polished by Newton's method in 50-digit arithmetic
              (oracle/kkt_mp.py: geo-mean pools through the n-asset KKT form, active set verified on the
              result) -- objective, psi, prices and EVERY pool's tenders to ~1e-30; this is what pins tenders
              at 1e-9 in the tests.  It showed the survey's per-pool vectors to be good to ~5e-6 only
              (arbitrage.py's constant-sum pool fills 0.3863495091 of its reserve, not 0.38634998);
  * "primal": the reference's primal model (arbitrage.py:51-78 etc.) solved by SciPy SLSQP
              (oracle/primal_scipy.py) -- the objective is good to ~1e-9, trades to ~1e-6;
  * "survey": the known answers of SURVEY.md Appendix B (derived in the survey session by SLSQP
              and by closed-form dual decomposition + L-BFGS-B, agreeing to >= 9 digits).
Neither is cvxpy output: cvxpy and every conic solver are absent from this container, so the
reference itself cannot be run here (PARITY UNPINNED by the reference; it holds no expected values).

  * "cvxpy":  THE REFERENCE'S OWN STACK, wherever it is importable: the three programs exactly as the scripts
              state them (tests/cvx_models.py: the same text the cfmm.cvx tests run) through `import cvxpy`,
              `prob.solve()` with cvxpy's default solver as in arbitrage.py:81-82 -- objective, psi, tenders,
              solver name and versions.  `python -B oracle/make_golden.py --cvxpy` adds / refreshes this key in
              the existing fixture without touching the other three; with cvxpy absent it says so and leaves the
              file alone.  tests/test_oracle.py::test_known_answers_against_the_reference_stack compares the
              other derivations (and, on the GPU, the HIP path through the same fixture) with it whenever the
              key exists -- the one route by which parity becomes pinned BY THE REFERENCE.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import instances as I          # noqa: E402
from oracle.primal_scipy import solve_primal   # noqa: E402
from oracle import kkt_mp                       # noqa: E402

# constant-sum pools that end on a kink ("PARTIAL fill"), with the direction of the kink
# (+1: tender the pool's first token).  kkt_mp.polish verifies the guess (fill inside (0,1), complementary slackness).
TIED = {"arbitrage": {4: -1}, "liquidation": {4: 1}, "two_asset_10": {4: -1}}

SURVEY = {
    "arbitrage": dict(value=21.4998087639, psi=[0, 1.174978045, 0, 3.2500094379],
                      nu=[3.01634, 10, 3.003003, 3],
                      y=[[-4.2335229132, 2.1355390109, -0.1310881388, 1.9283766914], [4.2335229132, -0.7363701675],
                         [-0.2241907984, 0.9134241780], [-4.6458358656, 5.1889999400], [3.8634998263, -3.8673671935]]),
    "liquidation": dict(value=15.8830108411, psi=[-2, -1, -3, -5, 15.8830108411],
                        nu=[0.2026491, 0.4646974, 0.3308995, 0.999, 1],
                        y=[[-7.2041541969, 0.0884031414, -0.1119111126, 3.0902355673, 3.5455728673],
                           [5.2041541969, -1.0884031414], [-2.8880888874, 3.7111490778],
                           [-4.6906436729, 5.2338077425], [-7.1107409722, 7.1036302313]]),
    "two_asset_0": dict(value=6.2330001314, nu=[1.0899570, 10.6037357, 1]),
    "two_asset_1": dict(value=7.3149009798, nu=[1.0294433, 10.2551254, 1]),
    "two_asset_10": dict(value=16.5943392089, nu=[1.0101010, 10.0210811, 1]),
    "two_asset_25": dict(value=31.4283406169, nu=[0.8189059, 8.9142302, 1]),
    "two_asset_49": dict(value=44.1820204014, nu=[0.3310097, 5.9262250, 1]),
}


def cases_all():
    cases = [("arbitrage", I.arbitrage()), ("liquidation", I.liquidation())]
    sweep = I.two_asset_sweep()
    for j in (0, 1, 10, 25, 49):
        cases.append((f"two_asset_{j}", I.two_asset(sweep[j])))
    return cases


def cvxpy_leg(inst):
    """the instance through the reference's own solver stack (arbitrage.py:39-84 as restated in tests/cvx_models.py);
    None when cvxpy is not importable"""
    try:
        import cvxpy
    except ImportError:
        return None
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.path.join(root, "tests") not in sys.path:
        sys.path.insert(0, os.path.join(root, "tests"))
    import cvx_models
    prob, goal, net, tender, receive = cvx_models.build(cvxpy, inst)
    prob.solve()                                   # (the scripts' call: default solver, default tolerances)
    stats = getattr(prob, "solver_stats", None)
    return dict(value=float(prob.value), status=str(prob.status), psi=np.asarray(net.value, float).tolist(),
                y=[(np.asarray(l.value, float) - np.asarray(d.value, float)).tolist() for d, l in zip(tender, receive)],
                solver=str(getattr(stats, "solver_name", "")), cvxpy_version=str(cvxpy.__version__))


def golden_path():
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "shipped_instances.json")


def add_cvxpy():
    """--cvxpy: add / refresh the "cvxpy" key of the committed fixture, nothing else"""
    try:
        import cvxpy  # noqa: F401
    except ImportError:
        print("cvxpy is not importable here: the fixture keeps its three derivations (parity unpinned by the reference)")
        return 1
    with open(golden_path()) as f:
        out = json.load(f)
    for name, inst in cases_all():
        out[name]["cvxpy"] = cvxpy_leg(inst)
        print(name, out[name]["cvxpy"]["value"], out[name]["kkt"]["value_str"], out[name]["cvxpy"]["solver"])
    with open(golden_path(), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", golden_path())
    return 0


def main():
    if "--cvxpy" in sys.argv[1:]:
        return add_cvxpy()
    out = {}
    for name, inst in cases_all():
        r = solve_primal(I.normalise(inst))
        k = kkt_mp.polish(I.normalise(inst), SURVEY[name]["nu"], TIED.get(name, {}))
        out[name] = dict(
            kkt=k,
            primal=dict(value=r["value"], psi=r["psi"].tolist(), y=[v.tolist() for v in r["y"]]),
            survey=SURVEY[name],
            t=inst["utility"].get("h", [0])[0] if inst["name"] == "two_asset" else None)
        cv = cvxpy_leg(inst)
        if cv is not None:
            out[name]["cvxpy"] = cv
        print(name, k["value_str"], r["value"], SURVEY[name]["value"])
    path = golden_path()
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    sys.exit(main() or 0)
