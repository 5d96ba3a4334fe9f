"""what a publisher -> followers hand-off inside one XCD costs (cfmm_time_xcd_handoff): the price of running iter_kernel's update on one
workgroup per XCD instead of on every workgroup.   python tools/xcd_handoff.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")]
import numpy as np, cfmm
from cfmm import synthetic
net = synthetic.config("C2")
p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
ctx = p._ensure_ctx()
ctx.time_xcd_handoff(2048, 20)                      # warm-up
rows = {f"{8 * npd} B": ctx.time_xcd_handoff(npd, 100) for npd in (128, 1024, 2048, 4096)}
print(json.dumps(dict(note="publisher: np doubles stored, vmcnt(0), barrier, flag (relaxed, agent scope); follower: spin on the flag of its own XCD (s_sleep 1), "
                           "load the doubles into LDS; 256 workgroups of 1024 threads, 8 publishers; us from the publisher's data-ready stamp "
                           "to the SLOWEST follower's data-in-LDS stamp (100 MHz wall clock); C3's trial prices + log-prices are 16 KB", rows=rows)))
p.close()
