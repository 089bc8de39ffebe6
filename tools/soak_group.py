"""Soak of the packed learners (oprl_group_step_n): N members of each algorithm / precision stepped for many updates, twice
from the same seeds: every parameter finite, no bounded wait expired, the two runs bit-identical.
``python tools/soak_group.py [scale]``"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch as t
import bench
from oprl_amd.algos.ddpg import DDPG
from oprl_amd.algos.sac import SAC
from oprl_amd.algos.td3 import TD3
from oprl_amd.group import LearnerGroup
from oprl_amd.logging import NullLogger

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
replay = bench.make_replay(t.device("cuda", 0), seed=7)
CASES = [("ddpg", DDPG, {}, "f32", 32, 20000), ("ddpg", DDPG, {}, "x2", 32, 10000), ("ddpg", DDPG, {}, "bf16", 16, 10000),
         ("td3", TD3, dict(log_every=10 ** 9), "x2", 32, 6000), ("td3", TD3, dict(log_every=10 ** 9), "f32", 8, 6000),
         ("sac", SAC, dict(log_every=10 ** 9, tune_alpha=True), "x2", 32, 5000), ("sac", SAC, dict(log_every=10 ** 9, tune_alpha=True), "bf16", 24, 5000)]
ok_all = True
for name, cls, kw, prec, n, updates in CASES:
    K = int(updates * scale)
    sums = []
    t0 = time.perf_counter()
    for rep in range(2):
        algos = []
        for i in range(n):
            t.manual_seed(500 + i)
            algos.append(cls(logger=NullLogger(), state_dim=bench.S, action_dim=bench.A, device="cuda:0", max_batch=256, precision=prec, **kw).create())
        g = LearnerGroup(algos)
        done = 0
        while done < K:
            k = min(997, K - done)          # (not a multiple of the chunk of four argument blocks, nor of TD3's policy_freq)
            k -= k % 2 if name == "td3" else 0
            g.step_n(replay.handle, k, 256, [1000 + i for i in range(n)])
            done += k
        t.cuda.synchronize()
        for a in algos:
            a.learner.check()
        finite = all(bool(t.isfinite(a.actor._oprl_arena).all()) and bool(t.isfinite(a.critic._oprl_arena).all()) for a in algos)
        sums.append(t.stack([a.critic._oprl_arena.double().sum() + a.actor._oprl_arena.double().sum() for a in algos]).cpu())
        g.close()
        del algos, g
    same = bool(t.equal(sums[0], sums[1]))
    ok_all = ok_all and finite and same
    print(f"[group {name} {prec}] {n} members x {K} updates x2  finite={finite}  runs identical={same}  ({time.perf_counter() - t0:.1f} s)", flush=True)
print("GROUP_SOAK_OK" if ok_all else "GROUP_SOAK_FAILED")
