"""What-if timing probe (needs a GPU): DDPG walker B = 256 through step_n with ONE cross-workgroup wait of k_ddpg_chain
counted as already satisfied (oprl_learner_debug_expire sites 101 ..: the numbers such a learner computes are WRONG —
it reads what the waited-for workgroup has not written yet — but every other wait, load and stage runs as usual, so
the change of the update's period is what that hand-over contributes to the critical cycle):
    101  role A's tail does not wait for role B's partial q
    102  the critic pass does not wait for the critic's tiles (ct_done)
    104  the critic's tiles do not wait for role A's seeds
    105  the actor's tiles do not wait for du
    106  the critic's tiles do not wait for role B's rows
usage: what_if.py [x2|f32] [site ...]     (one process per site is safest: an expired wait poisons the learner)"""
import os
import sys
import time
os.environ["OPRL_AMD_WHAT_IF"] = "1"      # (the library refuses the timing-experiment sites without it)
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch as t
import bench
from oprl_amd import _capi

prec = sys.argv[1] if len(sys.argv) > 1 else "x2"
sites = [int(a) for a in sys.argv[2:]] or [0]
dev = t.device("cuda", 0)
replay = bench.make_replay(dev, 0)
for site in sites:
    t.manual_seed(0)
    algo = bench._make_algo("DDPG", 24, 6, 256, {}, dev, prec)
    L = algo.learner
    L.step_n(replay.handle, 2000, 256, seed=0)
    t.cuda.synchronize()
    if site:
        _capi.check(L.lib.oprl_learner_debug_expire(L.handle, site))
    best = 1e9
    for _ in range(3):
        L.step_n(replay.handle, 5, 256, seed=0)
        t.cuda.synchronize()
        t0 = time.perf_counter()
        L.step_n(replay.handle, 2000, 256, seed=0)
        t.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{prec} what-if {site}: {best / 2000 * 1e6:.2f} us per update", flush=True)
    del algo, L
