"""is the launch tail systematic?  Per-workgroup tile-phase durations of the last full iter_kernel launch of several solves
(-DCFMM_PHASE_TIMERS build: CFMM_LIB=.../libcfmm_hip_timers.so): correlation between solves, and between launches of different
length (max_evals) inside a solve."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")]
import numpy as np, cfmm
from cfmm import synthetic
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
net = synthetic.config(cfg, seed=0)
p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
for _ in range(3): p.solve(tol=1e-6)
def last(max_evals):
    p.ctx.debug_timers()
    p.solve(tol=1e-6, max_evals=max_evals, method="lbfgs")
    _, _, tbi = p.ctx.debug_timers()
    st0 = tbi[:256, 0].min()
    return (tbi[:256, 0] - st0) * 0.01, (tbi[256:512, 0] - st0) * 0.01, (tbi[:256, 1] - st0) * 0.01
rows = {}
for me in (6, 6, 12, 12, 18, 18):
    st, up, en = last(me)
    rows.setdefault(me, []).append((en - up, en, up - st))
def c(a, b): return round(float(np.corrcoef(a, b)[0, 1]), 3)
out = {"config": cfg}
for me, r in rows.items():
    out["same_launch_%d" % me] = dict(corr_tiles=c(r[0][0], r[1][0]), corr_end=c(r[0][1], r[1][1]), corr_update=c(r[0][2], r[1][2]),
                                      tiles_mean=round(float(r[0][0].mean()), 2), tiles_std=round(float(r[0][0].std()), 2),
                                      end_max_minus_median=round(float(r[0][1].max() - np.median(r[0][1])), 2),
                                      end_max_minus_mean=round(float(r[0][1].max() - r[0][1].mean()), 2))
out["across_6_12"] = c(rows[6][0][0], rows[12][0][0]); out["across_12_18"] = c(rows[12][0][0], rows[18][0][0])
t = rows[12][0][0]
out["tiles_by_xcd_mean"] = [round(float(t[x::8].mean()), 2) for x in range(8)]
out["slowest_wgs"] = [int(i) for i in np.argsort(-t)[:12]]
out["fastest_wgs"] = [int(i) for i in np.argsort(t)[:12]]
print(json.dumps(out))
